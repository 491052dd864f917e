// comm_p2p.hip — one-shot all-reduce over peer-mapped HBM (xGMI), fused with the residual add and the RMSNorm that follow the
// row-parallel GEMMs of a tensor-parallel rank: RowParallelLinear::forward's exchange (reference src/layers/linear.rs:236-238,
// a TODO there) + Qwen3DecoderLayer's residual / norm (src/models/qwen3.rs:382-389; OptimizedRMSNorm::forward_with_residual,
// src/layers/layernorm.rs:170-176).  One launch replaces ncclAllReduce (>= 6 us of launch + a ring over 7 links for a 64 KB
// message) and the add+RMSNorm kernel; it is an ordinary kernel node, so the tensor-parallel decode step captures into a hipGraph.
//
// Every rank owns an ARENA in its own HBM, fine-grained and mapped into every peer (hipIpc handles across processes; plain
// pointers for the in-process test group):
//     slots[2 parities][8 source ranks][slot_bytes]   payload pushed BY the source rank
//     flags[2 parities][8 source ranks][PUSH_SPLIT]   epoch words, written by the source after its payload
// A collective with epoch e (parity e & 1) on rank me:
//   push workgroups (PUSH_SPLIT per peer): copy a quarter of my partial sums into peer.slots[parity][me] with system-scope
//     write-through stores, fence, barrier, then one lane stores e into peer.flags[parity][me][quarter] (release, system scope).
//     xGMI is point to point: the 7 peers are written over 7 links at once, nothing is forwarded.
//   reduce workgroups (one per row): poll MY flags of all peers until they read e (bounded: a peer that never arrives sets the
//     error word instead of hanging the GPU), acquire, then  y = fp16(sum over ranks in RANK ORDER, f32)  — my own rank's term
//     straight from my input — h <- fp16(h + y), out = rmsnorm(h) * w (or just y for the plain all-reduce).  Every rank adds
//     the same values in the same order: all ranks hold bit-identical results, equal to the oracle's tensor-parallel sum.
// Buffers alternate by parity: a peer can only be two collectives ahead of me after I have finished reading the slot it would
// overwrite (it needs my flag of the collective in between, which I send from a later kernel on my stream).
// The last workgroup to finish bumps the rank's epoch word (device memory: graph replays advance it without new arguments).
//
// Ordering without cache-maintenance fences (r05).  Every payload byte is written by a system-scope WRITE-THROUGH store (sc0 sc1) and read
// by a system-scope load (sc0 sc1): neither side keeps a copy in a non-coherent cache, so there is nothing for a release fence to write back
// (buffer_wbl2) or for an acquire fence to invalidate (buffer_inv).  What is left of release / acquire is ORDER: every storing wave waits for
// its stores to be acknowledged (s_waitcnt vmcnt(0)), the workgroup's barrier collects the waves, then one lane stores the flag (a system-scope
// store to the same peer, issued after the payload's acknowledgements); the reader polls the flag with system-scope loads, passes the
// workgroup barrier (a compiler barrier as well) and only then loads the payload (cdna guide Guideline 16, R1: "write-through, so no release
// fence"; sc1 loads in place of the acquire).  The epoch bump at the end is control flow only (the last workgroup to ARRIVE stores epoch + 1
// after every workgroup has read epoch): relaxed atomics.  r04 fenced each of these steps (__threadfence_system per push workgroup, a
// system-scope acquire per reduce workgroup, __threadfence + acq_rel per workgroup at the end): measured with a real peer on the same device
// (two in-process ranks, Qwen3-0.6B bs 32 x 1024, 57 collectives per step) 2.19 -> 1.95 ms/step, Qwen3-8B 7.37 -> 7.12
// (profiles/r05_tp_exchange.txt).
//
// The FENCED form (r06: P2PArgs::fenced, Comm::p2p_fenced, NVR_P2P_FENCED=1 / nvr_runner_p2p_set_fenced) is r04's protocol, compiled into the same
// kernels behind one uniform branch: __threadfence_system in front of a RELEASE flag store, a system-scope ACQUIRE fence behind the poll, release /
// acq_rel on the epoch words.  The fence-free form rests on "acknowledged write-through store => visible at the peer before the later flag store", which
// was measured on one device only (peers sharing HBM); until a run over real xGMI links has passed, the fenced form is what a failing self-test falls
// back to BEFORE RCCL (bench.py: fence-free -> fenced -> RCCL, recorded in config.collective_backend).
#include "kernels.h"
#include "device_utils.h"
#include "../common.h"
#include "../comm.h"

namespace nvr { namespace NVR_DT_NS {

// P2PArgs carries 16-bit payload pointers untyped (comm.h); this build reads them as its half_t (fp16, or bfloat16 with -DNVR_BF16)
#define HP(p) reinterpret_cast<half_t *>(p)
#define CHP(p) reinterpret_cast<const half_t *>(p)

__device__ __forceinline__ unsigned long long p2p_now() { return wall_clock64(); }   // s_memrealtime: 100 MHz, independent of the shader clock

// the last workgroup of a launch to get here stores epoch + 1 (every workgroup of the launch has read the epoch word by then); one lane per workgroup
__device__ __forceinline__ void p2p_advance_epoch(const P2PArgs &a, unsigned epoch) {
    if (a.fenced) {
        __threadfence();
        if (__hip_atomic_fetch_add(a.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
            __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.epoch, epoch + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (__hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
        __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.epoch, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

template <int P>   // P = 8 or 4 fp16 elements per thread and chunk
__global__ __launch_bounds__(256) void p2p_allreduce_kernel(P2PArgs a) {
    typedef half_t hp_t __attribute__((ext_vector_type(P)));
    const int tid = threadIdx.x;
    const unsigned epoch = *reinterpret_cast<volatile unsigned *>(a.epoch);
    const int parity = epoch & 1;
    const int npush = (a.nranks - 1) * P2P_PUSH_SPLIT;
    const size_t slot_elems = a.slot_bytes / 2;
    if ((int)blockIdx.x < npush) {
        // ---- push: my partial sums -> peer's slot of me
        const int pi = blockIdx.x / P2P_PUSH_SPLIT, sub = blockIdx.x % P2P_PUSH_SPLIT;
        const int peer = pi < a.rank ? pi : pi + 1;
        const size_t total16 = (a.count * 2 + 15) / 16;                        // 16-byte pieces of the payload
        const size_t per = (total16 + P2P_PUSH_SPLIT - 1) / P2P_PUSH_SPLIT, lo = sub * per, hi = min(total16, lo + per);
        half_t *dst = HP(a.peer_slots[peer]) + ((size_t)parity * 8 + a.rank) * slot_elems;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(total16 * 16), 0x00020000);
        typedef unsigned int u4 __attribute__((ext_vector_type(4)));
        // PB pieces requested per thread before the first store: as a plain load / store loop every iteration waits for its own load
        // (one round trip per 4 KiB of this workgroup's quarter: 4 for a 64 KiB message, 16 for 256 KiB)
        constexpr int PB = 8;
        for (size_t i0 = lo + tid; i0 < hi; i0 += 256 * PB) {
            u4 v[PB];
#pragma unroll
            for (int f = 0; f < PB; ++f) {
                const size_t i = i0 + (size_t)f * 256;
                v[f] = *reinterpret_cast<const u4 *>(reinterpret_cast<const char *>(a.in) + (i < hi ? i : hi - 1) * 16);
            }
#pragma unroll
            for (int f = 0; f < PB; ++f) {
                const size_t i = i0 + (size_t)f * 256;
                if (i < hi) __builtin_amdgcn_raw_buffer_store_b128(v[f], rs, (int)(i * 16), 0, 17);   // sc0 sc1: system scope, write-through
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      // every storing wave: its write-through stores are acknowledged
        if (a.fenced) __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            unsigned int *fl = a.peer_flags[peer] + ((size_t)parity * 8 + a.rank) * P2P_PUSH_SPLIT + sub;
            if (a.fenced) __hip_atomic_store(fl, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            else __hip_atomic_store(fl, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else {
        // ---- reduce row `row` of [rows][Hd]
        const int row = blockIdx.x - npush;
        __shared__ float sm[4];
        __shared__ int ok_s;
        if (tid < 64) {
            bool ok = true;
            if (tid < npush) {
                const int pi = tid / P2P_PUSH_SPLIT, sub = tid % P2P_PUSH_SPLIT;
                const int src = pi < a.rank ? pi : pi + 1;
                const unsigned *f = a.flags + ((size_t)parity * 8 + src) * P2P_PUSH_SPLIT + sub;
                const unsigned long long t0 = p2p_now();
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
                    __builtin_amdgcn_s_sleep(2);
                    if (p2p_now() - t0 > a.timeout_ticks) { ok = false; break; }
                }
            }
            ok = __all(ok);
            if (tid == 0) { ok_s = ok; if (!ok) __hip_atomic_store(a.err, epoch ? epoch : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (a.fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");          // system scope: the pushed payload is visible
        }
        __syncthreads();                                                         // (fence-free: every payload load below is a system-scope load)
        const bool ok = ok_s != 0;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(HP(a.slots) + (size_t)parity * 8 * slot_elems, 0, (int)(8 * a.slot_bytes), 0x00020000);
        constexpr int C = 4;                                                       // up to 4 chunks of 256*P elements per row
        hp_t v[C], g[C];
        float ss = 0.f;
        const size_t rbase = (size_t)row * a.Hd;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const int c = tid * P + i * (256 * P);
            if (c < a.Hd) {
                hp_t x[8];                                                         // all ranks' pieces requested together
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    if (r >= a.nranks || (r != a.rank && !ok)) { for (int j = 0; j < P; ++j) x[r][j] = (half_t)0.f; }
                    else if (r == a.rank) x[r] = *reinterpret_cast<const hp_t *>(CHP(a.in) + rbase + c);
                    else if constexpr (P == 8) {
                        x[r] = __builtin_bit_cast(hp_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(((size_t)r * slot_elems + rbase + c) * 2), 0, 17));
                    } else {
                        x[r] = __builtin_bit_cast(hp_t, __builtin_amdgcn_raw_buffer_load_b64(rs, (int)(((size_t)r * slot_elems + rbase + c) * 2), 0, 17));
                    }
                }
                if (a.gather_stride) {                                             // all-gather form: every rank's piece, unreduced
#pragma unroll
                    for (int r = 0; r < 8; ++r)
                        if (r < a.nranks) *reinterpret_cast<hp_t *>(HP(a.out) + (size_t)r * a.gather_stride + rbase + c) = x[r];
                    continue;
                }
                float acc[P];                                                      // rank order: the same sum on every rank
#pragma unroll
                for (int j = 0; j < P; ++j) acc[j] = (float)x[0][j];
#pragma unroll
                for (int r = 1; r < 8; ++r)
                    if (r < a.nranks) {
#pragma unroll
                        for (int j = 0; j < P; ++j) acc[j] += (float)x[r][j];
                    }
                hp_t y;
#pragma unroll
                for (int j = 0; j < P; ++j) y[j] = to_half_rn(acc[j]);
                if (a.h) {                                                         // residual add, layernorm.rs:170-176
                    v[i] = *reinterpret_cast<const hp_t *>(CHP(a.h) + rbase + c);
                    g[i] = *reinterpret_cast<const hp_t *>(CHP(a.wn) + c);
#pragma unroll
                    for (int j = 0; j < P; ++j) { v[i][j] = to_half_rn((float)v[i][j] + (float)y[j]); const float f = (float)v[i][j]; ss += f * f; }
                    *reinterpret_cast<hp_t *>(HP(a.h) + rbase + c) = v[i];
                } else {
                    *reinterpret_cast<hp_t *>(HP(a.out) + rbase + c) = y;              // plain all-reduce (in place allowed: own row only)
                }
            }
        }
        if (a.h) {
            ss = wave_sum(ss);
            if ((tid & 63) == 0) sm[tid >> 6] = ss;
            __syncthreads();
            const float rms = sqrtf((sm[0] + sm[1] + sm[2] + sm[3]) / (float)a.Hd + a.eps);
#pragma unroll
            for (int i = 0; i < C; ++i) {
                const int c = tid * P + i * (256 * P);
                if (c < a.Hd) {
                    hp_t o;
#pragma unroll
                    for (int j = 0; j < P; ++j) o[j] = to_half_rn(__fmul_rn(__fdiv_rn((float)v[i][j], rms), (float)g[i][j]));
                    *reinterpret_cast<hp_t *>(HP(a.out) + rbase + c) = o;
                }
            }
        }
    }
    // ---- the last workgroup to finish advances the epoch (every workgroup of the launch has read it by then)
    __syncthreads();
    if (tid == 0) p2p_advance_epoch(a, epoch);
}

// All-gather of a small per-rank record (greedy sampling under vocabulary sharding: (max, argmax) pairs, reference
// src/layers/embed_head.rs:321-336): one push workgroup per peer writes my record into peer.gslots[parity][me] and its flag;
// the last workgroup waits for every peer's flag and assembles recv[rank][bytes] in local memory.
__global__ __launch_bounds__(256) void p2p_allgather_kernel(P2PArgs a, const char *send, char *recv, int bytes) {
    const int tid = threadIdx.x;
    const unsigned epoch = *reinterpret_cast<volatile unsigned *>(a.epoch);
    const int parity = epoch & 1;
    const int npush = a.nranks - 1;
    if ((int)blockIdx.x < npush) {
        const int peer = (int)blockIdx.x < a.rank ? blockIdx.x : blockIdx.x + 1;
        char *dst = a.peer_gslots[peer] + ((size_t)parity * 8 + a.rank) * P2P_GATHER_BYTES;
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, P2P_GATHER_BYTES, 0x00020000);
        for (int i = tid * 4; i < bytes; i += 1024)
            __builtin_amdgcn_raw_buffer_store_b32(*reinterpret_cast<const unsigned *>(send + i), rs, i, 0, 17);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.fenced) __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            if (a.fenced) __hip_atomic_store(a.peer_gflags[peer] + parity * 8 + a.rank, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            else __hip_atomic_store(a.peer_gflags[peer] + parity * 8 + a.rank, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    } else {
        __shared__ int ok_s;
        if (tid < 64) {
            bool ok = true;
            if (tid < npush) {
                const int src = tid < a.rank ? tid : tid + 1;
                const unsigned *f = a.gflags + parity * 8 + src;
                const unsigned long long t0 = p2p_now();
                while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch) {
                    __builtin_amdgcn_s_sleep(2);
                    if (p2p_now() - t0 > a.timeout_ticks) { ok = false; break; }
                }
            }
            ok = __all(ok);
            if (tid == 0) { ok_s = ok; if (!ok) __hip_atomic_store(a.err, epoch ? epoch : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
            if (a.fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        }
        __syncthreads();
        const auto rs = __builtin_amdgcn_make_buffer_rsrc(a.gslots + (size_t)parity * 8 * P2P_GATHER_BYTES, 0, 8 * P2P_GATHER_BYTES, 0x00020000);
        for (int r = 0; r < a.nranks; ++r)
            for (int i = tid * 4; i < bytes; i += 1024) {
                unsigned v;
                if (r == a.rank) v = *reinterpret_cast<const unsigned *>(send + i);
                else v = ok_s ? __builtin_amdgcn_raw_buffer_load_b32(rs, r * P2P_GATHER_BYTES + i, 0, 17) : 0u;
                *reinterpret_cast<unsigned *>(recv + (size_t)r * bytes + i) = v;
            }
    }
    __syncthreads();
    if (tid == 0) p2p_advance_epoch(a, epoch);
}

int p2p_allgather_launch(const P2PArgs &a, const void *send, void *recv, size_t bytes, hipStream_t s) {
    if (a.nranks < 2 || a.nranks > 8) return fail(NVR_ERR_INVALID_ARG, "p2p all-gather: %d ranks (2..8)", a.nranks);
    if (bytes == 0 || bytes % 4 || bytes > (size_t)P2P_GATHER_BYTES) return fail(NVR_ERR_UNSUPPORTED, "p2p all-gather: %zu bytes per rank (multiple of 4, <= %d)", bytes, P2P_GATHER_BYTES);
    p2p_allgather_kernel<<<dim3((unsigned)a.nranks), dim3(256), 0, s>>>(a, (const char *)send, (char *)recv, (int)bytes);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NVR_ERR_HIP, "p2p all-gather launch failed: %s", hipGetErrorString(e));
    return NVR_OK;
}

int p2p_allreduce_launch(const P2PArgs &a, int rows, hipStream_t s) {
    if (a.nranks < 2 || a.nranks > 8) return fail(NVR_ERR_INVALID_ARG, "p2p all-reduce: %d ranks (2..8)", a.nranks);
    if (a.Hd % 4 || a.Hd > 8192 || (a.Hd % 8 && a.Hd > 4096) || (size_t)rows * a.Hd != a.count || a.count * 2 > a.slot_bytes)
        return fail(NVR_ERR_UNSUPPORTED, "p2p all-reduce: %d rows x %d (<= 8192, multiple of 4), %zu bytes per slot", rows, a.Hd, a.slot_bytes);
    if ((a.nranks - 1) * P2P_PUSH_SPLIT > 64) return fail(NVR_ERR_INVARIANT, "p2p all-reduce: too many push flags for one wave");
    const unsigned grid = (unsigned)((a.nranks - 1) * P2P_PUSH_SPLIT + rows);
    if (a.Hd % 8 == 0 && a.Hd > 1024) p2p_allreduce_kernel<8><<<dim3(grid), dim3(256), 0, s>>>(a);
    else p2p_allreduce_kernel<4><<<dim3(grid), dim3(256), 0, s>>>(a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(NVR_ERR_HIP, "p2p all-reduce launch failed: %s", hipGetErrorString(e));
    return NVR_OK;
}

}}  // namespace nvr::NVR_DT_NS
