// device_common.h -- device helpers shared by the kernel translation units (gfx950 only).
#pragma once
#include "kernels.h"

// Diagnostic hooks: empty in the library.  tools/archive/phase_probe.hip defines them before including a kernel source to
// record in-kernel time stamps (cdna_hip_programming.md section 7, "In-kernel stamps": a separate diagnostic build, the
// stamps go to a buffer of their own).
#ifndef FWA_STAMP
#define FWA_STAMP(slot)
#endif
#ifndef FWA_STAMP_B  // the last-pass kernels, so that a probe can tell two concurrently running kernels apart
#define FWA_STAMP_B(slot) FWA_STAMP(slot)
#endif
#ifndef FWA_ENTRY_HOOK
#define FWA_ENTRY_HOOK()
#endif

namespace fwa {

// Buffer (SRD) addressing: one 32-bit per-lane byte offset + a scalar offset per access, so the 32 loads
// and 32 stores of a tile need no per-access VALU address math (cdna_hip_programming.md T8/T20).  The
// descriptor covers exactly one 8-MiB transform; out-of-range lanes would read 0 / drop the store.
typedef unsigned v2u __attribute__((ext_vector_type(2)));
constexpr uint32_t TRANSFORM_BYTES = 8u << 20;
// cache-policy bits of the aux operand (gfx940+): sc0 = 1, nt = 2, sc1 = 16
constexpr int AUX_DEFAULT = 0, AUX_NT = 2, AUX_SC1 = 16;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const v2f *transform_base)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<v2f *>(transform_base), 0, TRANSFORM_BYTES, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ v2f buf_load(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    return __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, AUX));
}
template <int AUX>
__device__ __forceinline__ void buf_store(v2f v, __amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff)
{
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), r, voff, soff, AUX);
}


// XCD-aware block -> tile mapping shared by the pass kernels (cdna_hip_programming.md T1).  Blocks are dealt round-robin over the
// 8 XCDs.  bit 0: every XCD takes a contiguous run of tile indices instead of one residue class mod 8 (its resident workgroups
// then cover whole rows).  bit 2 (with bit 0; round 4): inside every run of 64, the i-th and (i + 32)-th workgroup of the XCD --
// the two residents of one CU when the kernel runs two workgroups per CU and has the chip to itself -- take ADJACENT tiles: their
// strided pieces are the two halves of a piece twice as long (profiles/round4/sweep_pair_map_cu_split_negative.txt).  Needs a
// grid that is a multiple of 8 (launchers clear the bits otherwise) and, for bit 2, whole runs of 64 per XCD.
__device__ __forceinline__ uint32_t xcd_map(uint32_t swizzle)
{
    const uint32_t b = blockIdx.x;
    if (!(swizzle & 1u)) return b;
    const uint32_t per = gridDim.x >> 3;
    uint32_t i = b >> 3;
    if ((swizzle & 4u) && (per & 63u) == 0) {
        const uint32_t r = i & 63u;
        i = (i & ~63u) + (((r & 31u) << 1) | (r >> 5));
    }
    return (b & 7u) * per + i;
}

// Workgroup -> chunk index of the one-launch (streaming) kernels k_chunk, k_small32, k_scale, k_copy (round 5).  Blocks are
// dealt round-robin over the 8 XCDs; with the plain map XCD x works on every eighth 64-KiB chunk of one moving window.  Map 5
// (xcd_map bits 0 + 2: every XCD a contiguous run of chunks, and the i-th and (i + 32)-th workgroup of an XCD -- two residents
// of one CU -- on ADJACENT chunks) is worth + 1.3 ... + 7.6 % at every size from 2 to 32768 and + 2 ... + 8 % on normalize at
// the 32-GiB footprint, one library per map side by side in one process (tools/ab_libs.py,
// profiles/round5/ab_one_launch_block_map.jsonl; a pure streaming pass: + 1.6-2.4 %, probe_stream_shapes_32GiB.txt).  Bit 0
// alone LOSES up to 11 % at some sizes, four or eight adjacent chunks per CU lose 1-9 %
// (ab_one_launch_block_map_pairs_quads_octs_16GiB.jsonl); the XCDs' runs interleaved in pieces of 2 or 64 chunks instead of one
// contiguous run each: level with it within the +- 1.5 % run-to-run spread (ab_one_launch_block_map_interleaved_runs.jsonl).
// Results do not depend on the map.
#ifndef FWA_ONE_LAUNCH_MAP
#define FWA_ONE_LAUNCH_MAP 5
#endif
__device__ __forceinline__ uint32_t one_launch_block()
{
    if constexpr (FWA_ONE_LAUNCH_MAP == 0) return blockIdx.x;
    // the map covers the largest prefix of the grid that is whole runs of 64 workgroups per XCD (a multiple of 512 blocks:
    // bit 0 without bit 2 is the map that loses); the ragged remainder keeps the plain map
    const uint32_t b = blockIdx.x, whole = gridDim.x & ~511u;
    if (b >= whole) return b;
    const uint32_t per = whole >> 3;
    uint32_t i = b >> 3;
    if constexpr ((FWA_ONE_LAUNCH_MAP & 4) != 0) {
        const uint32_t r = i & 63u;
        i = (i & ~63u) + (((r & 31u) << 1) | (r >> 5));
    }
    return (b & 7u) * per + i;
}

template <int N>
__device__ __forceinline__ v2f tw_lookup(const v2f *__restrict__ tw, uint32_t e)  // W_N^e, 0 <= e < N
{
    const v2f w = tw[e & (N / 2 - 1)];
    return (e & (N / 2)) ? -w : w;
}

template <int R, int N, int DIR, class Get, class Put>
__device__ __forceinline__ void stage_bfly(Get get, Put put, const v2f *__restrict__ tw, uint32_t idx, uint32_t J)
{
    // one radix-R butterfly of the Stockham stage with sub-block size J
    v2f x[R];
    static_for<0, R>([&](auto m_) { constexpr int m = decltype(m_)::value; x[m] = get(idx + m * (N / R)); });
    fft_reg<R, DIR>(x);
    const uint32_t j = idx & (J - 1);
    const uint32_t sJ = idx - j;  // s*J
    static_for<0, R>([&](auto q_) {
        constexpr int q = decltype(q_)::value;
        v2f v = x[brev<R>(q)];
        if constexpr (q != 0) {
            if (J * R < N) v = cmul_tw<DIR>(v, tw_lookup<N>(tw, sJ * q));  // last stage: s = 0, no twiddle
        }
        put(sJ * R + j + q * J, v);
    });
}

}  // namespace fwa
