// Hand-off latency between two workgroups, by placement (same XCD / other XCD) and store flavour
// (write-through sc1 store vs plain store that stays in the producer XCD's L2), consumer always polls with sc1 loads
// (L1 bypass).  Round trip = A publishes i, P sees it and answers i, A sees the answer.  One 256-thread workgroup per
// CU (64 KB of LDS each), 256 workgroups; workgroup 0 is A.  hipcc --offload-arch=gfx950 tools/xcd_handoff.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define N_IT 2000
#define LIMIT (1 << 22)
__device__ __forceinline__ unsigned ld_sc1(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_sc1(unsigned *p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_plain(unsigned *p, unsigned v) {
    asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
// buf: [0..255] xcc ids (0xffffffff until written) | [256] partner same | [257] partner other | [300 + 64 j] slots
__global__ __launch_bounds__(256) void k(unsigned *buf, unsigned long long *out, int plain) {
    extern __shared__ float lds[];
    const int w = blockIdx.x, tid = threadIdx.x;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (tid == 0) st_sc1(buf + w, xcc);
    __shared__ int role, partner, err;
    if (tid == 0) {
        role = -1; partner = -1; err = 0;
        if (w == 0) {
            int same = -1, other = -1;
            for (int spins = 0; spins < LIMIT && (same < 0 || other < 0); ++spins)
                for (int i = 1; i < (int)gridDim.x; ++i) {
                    const unsigned x = ld_sc1(buf + i);
                    if (x == 0xffffffffu) continue;
                    if (x == xcc && same < 0) same = i;
                    if (x != xcc && other < 0) other = i;
                }
            st_sc1(buf + 256, (unsigned)same);
            st_sc1(buf + 257, (unsigned)other);
            st_sc1(buf + 258, 1u);
        }
        for (int spins = 0; spins < LIMIT && ld_sc1(buf + 258) != 1u; ++spins) __builtin_amdgcn_s_sleep(8);
        const int same = (int)ld_sc1(buf + 256), other = (int)ld_sc1(buf + 257);
        if (w == 0) role = 0;
        else if (w == same) role = 1;
        else if (w == other) role = 2;
        partner = w == 0 ? same : 0;
        if (w == 0) { out[4] = xcc; out[5] = same; out[6] = other; }
    }
    __syncthreads();
    if (role < 0 || tid != 0) return;
    for (int phase = 1; phase <= 2; ++phase) {
        unsigned *sa = buf + 320 + 64 * (2 * phase), *sp = buf + 320 + 64 * (2 * phase + 1);
        if (role == 0) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            for (unsigned i = 1; i <= N_IT; ++i) {
                if (plain) st_plain(sa, i); else st_sc1(sa, i);
                int spins = 0;
                while (ld_sc1(sp) != i && ++spins < LIMIT) {}
                if (spins >= LIMIT) { out[7] = phase; return; }
            }
            out[phase - 1] = __builtin_amdgcn_s_memrealtime() - t0;
        } else if (role == phase) {
            for (unsigned i = 1; i <= N_IT; ++i) {
                int spins = 0;
                while (ld_sc1(sa) != i && ++spins < LIMIT) {}
                if (spins >= LIMIT) return;
                if (plain) st_plain(sp, i); else st_sc1(sp, i);
            }
        }
    }
}
int main() {
    unsigned *buf; unsigned long long *out;
    hipMalloc(&buf, 4096 * 4); hipMalloc(&out, 64);
    for (int plain = 0; plain <= 1; ++plain) {
        hipMemset(buf, 0xFF, 320 * 4); hipMemset(buf + 256, 0, 64 * 4 + 3000 * 4); hipMemset(out, 0, 64);
        hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        hipLaunchKernelGGL(k, dim3(256), dim3(256), 96 * 1024, 0, buf, out, plain);
        hipError_t e = hipDeviceSynchronize();
        unsigned long long h[8];
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("%s stores: A on XCD %llu; same-XCD partner wg %llu: %.3f us per round trip; other-XCD partner wg %llu: %.3f us per round trip%s (%s)\n",
               plain ? "plain" : "sc1  ", h[4], h[5], h[0] * 0.01 / N_IT, h[6], h[1] * 0.01 / N_IT,
               h[7] ? "  [TIMED OUT in this phase]" : "", hipGetErrorString(e));
    }
    return 0;
}
