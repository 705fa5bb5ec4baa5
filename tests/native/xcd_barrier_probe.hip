// xcd_barrier_probe.hip -- what a barrier-separated exchange stage costs (a) across the whole device with agent-scope (sc1) traffic, as
// gnn_persistent_kernel does it today, and (b) among the 32 workgroups of ONE XCD (blockIdx % 8 == 0) with L2-local traffic: stores plain
// (the vector L1 is write-through) and drained, loads with sc0 (miss the L1, hit the XCD's L2), counters by workgroup-scope atomics
// (executed in that L2).  Every round each worker writes a 256-float block that depends on the round, passes the barrier, reads ALL
// workers' blocks and checks them (a stale line would show).   hipcc --offload-arch=gfx950 -O3 ... && /tmp/xcd_barrier_probe
#include <cstdio>
#include <vector>
#include <hip/hip_runtime.h>
constexpr int SC0 = 1, SC1 = 16;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf(const void* p) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x7fffffff, 0x00020000); }
template <int POL> __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, int i) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, i * 4, 0, POL)); }
template <int POL> __device__ __forceinline__ void st(__amdgpu_buffer_rsrc_t r, int i, float v) { __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, i * 4, 0, POL); }

// LOCAL: one XCD.  workers = the workgroups that take part (all of them when !LOCAL)
template <bool LOCAL, int VAR>
__global__ __launch_bounds__(256) void k(float* data, unsigned* bar, int rounds, int workers, unsigned long long* ticks, unsigned* errors, int heavy) {
    if (LOCAL && (blockIdx.x & 7) != 0) return;
    const int me = LOCAL ? blockIdx.x >> 3 : blockIdx.x;
    float sink = 0.0f;
    const __amdgpu_buffer_rsrc_t db = buf(data);
    __shared__ int dummy;
    unsigned bad = 0;
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; r++) {
        float* mine = data + ((size_t)(r & 1) * workers + me) * 256;
        if (LOCAL) {
            if (VAR == 3) st<SC0>(db, (int)(mine - data) + threadIdx.x, (float)(r * 1000 + me) + threadIdx.x * 0.001f);
            else mine[threadIdx.x] = (float)(r * 1000 + me) + threadIdx.x * 0.001f;
        }
        else st<SC1>(db, (int)(mine - data) + threadIdx.x, (float)(r * 1000 + me) + threadIdx.x * 0.001f);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned target = (unsigned)(r + 1) * workers;
            if (LOCAL) {
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                for (int spin = 0; spin < 20000; spin++) {      // bounded: a stale line must not hang the device
                    unsigned v;
                    asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(bar) : "memory");
                    if (v >= target) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (spin == 19999) atomicAdd(errors + 1, 1u);
                }
            } else {
                __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int spin = 0; spin < 20000 && __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; spin++) {
                    __builtin_amdgcn_s_sleep(1);
                    if (spin == 19999) atomicAdd(errors + 1, 1u);
                }
            }
            dummy = 0;
        }
        __syncthreads();
        if (LOCAL && VAR == 2) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
        if (heavy) {
            float4 a[16];
#pragma unroll
            for (int t = 0; t < 16; t++) a[t] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(db, ((((r & 1) * workers + ((me + t) % workers)) * 256) + (threadIdx.x & 63) * 4) * 4, 0, SC1));
#pragma unroll
            for (int t = 0; t < 16; t++) sink += a[t].x + a[t].w;
        }
        // read one float of every worker's block of this round (lane -> worker), all of a few blocks
        for (int w = threadIdx.x; w < workers; w += 256) {
            const int idx = ((r & 1) * workers + w) * 256 + (threadIdx.x & 255);
            float v;
            if (!LOCAL) v = ld<SC1>(db, idx);
            else if (VAR == 0 || VAR == 3) v = ld<SC0>(db, idx);
            else if (VAR == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(data + idx) : "memory");
            else if (VAR == 2) v = data[idx];
            else if (VAR == 5) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(data + idx) : "memory");
            else asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(data + idx) : "memory");
            const float want = (float)(r * 1000 + w) + (threadIdx.x & 255) * 0.001f;
            if (v != want) bad++;
        }
    }
    if (me == 0 && threadIdx.x == 0) *ticks = wall_clock64() - t0;
    if (sink == 12345.678f) data[0] = sink;
    if (bad) atomicAdd(errors, bad);
}
int main() {
    float* data; unsigned* bar; unsigned long long* ticks; unsigned* err;
    (void)hipMalloc(&data, 2 * 256 * 256 * 4); (void)hipMalloc(&bar, 256); (void)hipMalloc(&ticks, 8); (void)hipMalloc(&err, 8);
    const int rounds = 500;
    for (int mode = 0; mode < 8; mode++) for (int rep = 0; rep < 1; rep++) {
        (void)hipMemset(bar, 0, 256); (void)hipMemset(err, 0, 8); (void)hipMemset(data, 0, 2 * 256 * 256 * 4);
        const int heavy = mode >= 4;
        if (mode == 0 || mode == 4) k<false, 0><<<256, 256>>>(data, bar, rounds, 256, ticks, err, heavy);
        else if (mode == 1 || mode == 5) k<true, 4><<<256, 256>>>(data, bar, rounds, 32, ticks, err, heavy);       // sc0 sc1 loads
        else if (mode == 2 || mode == 6) k<true, 5><<<256, 256>>>(data, bar, rounds, 32, ticks, err, heavy);       // sc1 loads
        else k<true, 5><<<512, 256>>>(data, bar, rounds, 64, ticks, err, heavy);                                    // 64 workers (two per CU), sc1 loads
        (void)hipDeviceSynchronize();
        unsigned long long t; unsigned e[2];
        (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(e, err, 8, hipMemcpyDeviceToHost);
        printf("mode %d heavy %d %s: %.2f us per round, %u wrong values, %u barrier time-outs\n", mode, heavy, (mode & 3) ? ((mode & 3) == 1 ? "one XCD, 32 workgroups, loads sc0 sc1" : (mode & 3) == 2 ? "one XCD, 32 workgroups, loads sc1" : "one XCD, 64 workgroups, loads sc1") : "whole device, 256 workgroups, sc1 / agent scope", t / 100.0 / rounds, e[0], e[1]); fflush(stdout);
    }
}
