// What a cross-stream dependency costs on the producer's queue: kernel A (writes N MB) -> hipEventRecord -> kernel B on one
// stream, the event created with different flags; a second stream waits for the event and runs a small kernel.  Reports
// the time from A's start to B's end per iteration (HIP events around the pair, default stream ordering).
// hipcc --offload-arch=gfx950 tools/event_cost.hip -o tools/bin/event_cost
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void fill(float4 *p, size_t n, float v) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(v, v, v, v);
}
__global__ void tiny(float *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.f; }
int main() {
    const size_t mb = 32, n = mb * 1024 * 1024 / 16;
    float4 *buf; float *small;
    hipMalloc(&buf, n * 16); hipMalloc(&small, 4096);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    hipEvent_t t0, t1; hipEventCreate(&t0); hipEventCreate(&t1);
    struct { const char *name; int mode; unsigned flags; } cases[] = {
        {"no event", 0, 0},
        {"record only, hipEventDisableTiming", 1, hipEventDisableTiming},
        {"record + other stream waits, hipEventDisableTiming", 2, hipEventDisableTiming},
        {"record + wait, DisableTiming|ReleaseToDevice", 2, hipEventDisableTiming | hipEventReleaseToDevice},
        {"record + wait, DisableTiming|ReleaseToSystem", 2, hipEventDisableTiming | hipEventReleaseToSystem},
        {"record + wait, default flags (timing on)", 2, hipEventDefault},
    };
    for (auto &c : cases) {
        hipEvent_t ev = nullptr;
        if (c.mode) hipEventCreateWithFlags(&ev, c.flags);
        float best = 1e9f, sum = 0.f;
        const int iters = 50;
        for (int it = 0; it < iters + 5; ++it) {
            hipEventRecord(t0, s1);
            hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, s1, buf, n, (float)it);
            if (c.mode >= 1) hipEventRecord(ev, s1);
            if (c.mode >= 2) { hipStreamWaitEvent(s2, ev, 0); hipLaunchKernelGGL(tiny, dim3(8), dim3(64), 0, s2, small); }
            hipLaunchKernelGGL(tiny, dim3(8), dim3(64), 0, s1, small + 64);
            hipEventRecord(t1, s1);
            hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, t0, t1);
            if (it >= 5) { sum += ms; best = ms < best ? ms : best; }
        }
        printf("%-52s  A(%zu MB)+B: mean %.1f us, min %.1f us\n", c.name, mb, sum / iters * 1e3, best * 1e3);
        if (ev) hipEventDestroy(ev);
    }
    return 0;
}
