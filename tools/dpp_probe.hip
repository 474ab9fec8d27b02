// What the DPP controls used by the scans do on gfx950 (run on the GPU: hipcc --offload-arch=gfx950 tools/dpp_probe.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int v = threadIdx.x;
    int r[6];
    r[0] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
    r[1] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    r[2] = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    r[3] = __builtin_amdgcn_update_dpp(-1, v, 0x150, 0xf, 0xf, false);  // row_newbcast:0 (gfx90a+): lane 0 of each row
    r[4] = __builtin_amdgcn_update_dpp(-7, v, 0x102, 0xf, 0xf, false);  // row_shl:2, lanes without a source keep `old`
    r[5] = __builtin_amdgcn_update_dpp(-7, v, 0x112, 0xf, 0xf, true);   // row_shr:2 with bound_ctrl: 0 where no source
    for (int i = 0; i < 6; i++) out[64 * i + threadIdx.x] = r[i];
}
int main() {
    int* d; hipMalloc(&d, 6 * 64 * 4);
    k<<<1, 64>>>(d);
    int h[6 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[6] = {"wave_shr:1", "wave_shl:1", "row_bcast:15 mask 0xa", "row_newbcast:0", "row_shl:2 keep old(-7)", "row_shr:2 bound_ctrl"};
    for (int r = 0; r < 6; r++) { printf("%-24s", names[r]); for (int i = 0; i < 64; i++) printf("%d ", h[r * 64 + i]); printf("\n"); }
    return 0;
}
