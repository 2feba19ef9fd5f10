// odpd_xchg.h — the one-shot gradient exchange of the data-parallel step ("one-shot direct/LL over ring" for a
// 2-12 KB message).  The step's only collective is a sum of P + 4 floats (~4 KB): a ring or tree all-reduce pays several launch +
// hop latencies for it; here every rank WRITES its vector straight into a slot row it owns inside every peer's memory (peer HBM
// mapped through hipIpc over xGMI, or host shared memory), then sums the rows that arrived in its own slots in rank order — one
// hop, no second kernel when folded into the optimiser kernel's prologue (optim.hip), and the fixed summation order makes the
// replicas bit-identical.
//
// Wire format ("LL": data and flag travel in one 8-byte word, so no fence orders them): word = (sequence number << 32) | fp32 bits,
// stored / polled with relaxed system-scope 8-byte atomics.  Slots of one rank: [2 parities][world source rows][row_stride words];
// step s uses parity s & 1.  Two parities suffice: a rank can only overwrite parity p (step s + 2) after it finished step s + 1,
// which needed every peer's step-(s + 1) row, which a peer only sends after its own step-s kernel (the reader of parity p) retired.
#pragma once
#include <hip/hip_runtime.h>

namespace odpd {

constexpr int kXchgMaxWorld = 8;          // one node
constexpr int kXchgMaxFloats = 8192;      // row_stride: the largest vector one exchange carries

struct XchgDev {                          // one exchange, by value into the kernel
    unsigned long long* dst[kXchgMaxWorld];   // dst[r]: this rank's row of this parity inside rank r's slots (dst[rank] unused)
    const unsigned long long* src;            // this rank's slots of this parity: row r = what rank r sent
    int* err;                                 // this rank's time-out counter (device memory)
    long long timeout_ticks;                  // of wall_clock64() (100 MHz): a peer that never arrives poisons the sum with NaN
    unsigned seq;                             // > 0
    int world, rank, row_stride;
};

// in-place g[0..n) = sum over ranks, by ONE workgroup (every thread of the block calls it; ends with a barrier)
__device__ __forceinline__ void xchg_allreduce_block(const XchgDev& xd, float* __restrict__ g, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const unsigned long long w = ((unsigned long long)xd.seq << 32) | (unsigned long long)__float_as_uint(g[i]);
#pragma unroll
        for (int r = 0; r < kXchgMaxWorld; ++r)
            if (r < xd.world && r != xd.rank) __hip_atomic_store(xd.dst[r] + i, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    bool any_late = false;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float acc = 0.f;
        bool late = false;
        const long long t0 = wall_clock64();
        for (int r = 0; r < xd.world; ++r) {          // rank order, on every rank: the replicas stay bit-identical
            float v = g[i];
            if (r != xd.rank) {
                const unsigned long long* p = xd.src + (size_t)r * xd.row_stride + i;
                unsigned long long w;
                while ((unsigned)((w = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) >> 32) != xd.seq) {
                    if (wall_clock64() - t0 > xd.timeout_ticks) { late = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                v = __uint_as_float((unsigned)w);
            }
            acc += v;
        }
        if (late) { acc = __uint_as_float(0x7fc00000u); any_late = true; }
        g[i] = acc;
    }
    // the counter counts EXCHANGES that timed out, not elements: one increment per call, by thread 0 after a block-wide OR
    if (__syncthreads_or(any_late) && threadIdx.x == 0) atomicAdd(xd.err, 1);
    __syncthreads();
}

}  // namespace odpd
