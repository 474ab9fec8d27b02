// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Lets the reference's single native source, /root/reference/source/kernels.cu, be parsed by g++
// as host C++ so that its kernels can be executed on a CPU by a host grid loop (SURVEY.md §8(c),
// §11).  kernels.cu includes nothing but <stdio.h>; the definitions below only give meaning to the
// CUDA *language keywords/builtins* it uses (no header, library or generated file is substituted).
// The reference file is read where it lies (path passed on the compiler command line by
// oracle/Makefile); nothing of it is copied into this repository.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <barrier>

using std::abs;
using std::max;
using std::min;

#define __global__
#define __device__

struct ref_dim3 {
    int x, y, z;
};
static thread_local ref_dim3 threadIdx, blockIdx;  // set per emulated thread
static ref_dim3 blockDim, gridDim;                 // set by the launcher

// __syncthreads(): no-op for the data-parallel kernels (they are run one emulated thread at a
// time); a real barrier when the launcher runs one std::thread per CUDA thread of the block
// (integrate_flux_double, whose phases are separated by block barriers).
static std::barrier<>* ref_block_barrier = nullptr;
static inline void __syncthreads() {
    if (ref_block_barrier) ref_block_barrier->arrive_and_wait();
}

static inline unsigned long long atomicCAS(unsigned long long* a, unsigned long long cmp,
                                           unsigned long long val) {
    __atomic_compare_exchange_n(a, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST);
    return cmp;
}
static inline float atomicExch(float* a, float v) {  // single-precision path, unused (fp64 only)
    float o = *a;
    *a = v;
    return o;
}
static inline long long __double_as_longlong(double d) {
    long long r;
    memcpy(&r, &d, 8);
    return r;
}
static inline double __longlong_as_double(long long l) {
    double r;
    memcpy(&r, &l, 8);
    return r;
}
