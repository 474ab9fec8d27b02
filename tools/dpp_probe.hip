#include <hip/hip_runtime.h>
__global__ void k(int* out) {
    int v = threadIdx.x;
    int a = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);  // wave_shr:1
    int b = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);  // wave_shl:1
    int c = __builtin_amdgcn_update_dpp(-1, v, 0x142, 0xa, 0xf, false);  // row_bcast:15
    out[threadIdx.x] = a; out[64 + threadIdx.x] = b; out[128 + threadIdx.x] = c;
}
int main() {
    int* d; hipMalloc(&d, 192 * 4);
    k<<<1, 64>>>(d);
    int h[192]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int r = 0; r < 3; r++) { for (int i = 0; i < 64; i++) printf("%d ", h[r * 64 + i]); printf("\n"); }
    return 0;
}
