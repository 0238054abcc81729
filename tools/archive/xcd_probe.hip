// tools/xcd_probe.hip -- measurement tool (not product code): could a 2^20-point transform live inside ONE XCD?
//
// The two-pass pipeline moves every sample across the L2<->fabric boundary four times (HBM in, ring out, ring
// in, HBM out).  The only way to drop the two ring crossings is to keep the 8-MiB intermediate inside an XCD:
// 64 workgroups (2 per CU x 32 CUs) hold the transform in registers (128 KiB each) and do the 1024x1024
// transpose between the passes as pairwise block swaps through a small L2-resident mailbox.  This probe times
// the memory/synchronisation skeleton of that design (no arithmetic):
//   part A  HBM rate of k = 1, 2, 4, 8 active XCDs (does one XCD alone get more than 1/8 of the chip?)
//   part B  the in-XCD exchange: R rounds of K = 32/R registers per thread, per round every workgroup stores K
//           blocks of 4 KiB to its partners' mailboxes, one XCD-wide barrier (a counter polled by one lane per
//           workgroup), then loads K blocks.  Pairing (a + b) mod 32 == phase keeps register indices static.
//           Variants: barrier counter with agent-scope atomics (memory side) or L2 atomics (same-XCD only).
//   part C  load tile from HBM -> exchange -> store tile to HBM per transform: the whole skeleton.
// Groups are formed from HW_REG_XCC_ID at run time; a launch whose XCDs do not each hold exactly 64 workgroups
// reports it and measures nothing.  Every spin is bounded (200 ms) and sets an error word.
//   hipcc --offload-arch=gfx950 -O3 -o tools/xcd_probe tools/xcd_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NT = 2, SC1 = 16;

__device__ __forceinline__ uint32_t xcc_id()
{
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}

// ---------------------------------------------------------------- part A
// mode 0 read, 1 write, 2 copy; blocks on XCDs >= k_active exit; the rest pull 256-KiB chunks from one counter
__global__ __launch_bounds__(256) void k_xcd_stream(const v4u *in, v4u *out, uint64_t n_chunks, uint32_t k_active, int mode,
                                                    uint32_t *counter, uint32_t *sink)
{
    __shared__ uint32_t s_chunk;
    if (xcc_id() >= k_active) return;
    v4u acc = {0, 0, 0, 0};
    for (;;) {
        if (threadIdx.x == 0) s_chunk = atomicAdd(counter, 1u);
        __syncthreads();
        const uint64_t c = s_chunk;
        __syncthreads();
        if (c >= n_chunks) break;
        const uint64_t base = c * 16384 + threadIdx.x;  // 16384 x 16 B = 256 KiB per chunk
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 64; i += 8) {
                v4u t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = __builtin_nontemporal_load(&in[base + (i + j) * 256]);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc ^= t[j];
            }
        } else if (mode == 1) {
#pragma unroll
            for (int i = 0; i < 64; ++i) __builtin_nontemporal_store(acc, &out[base + i * 256]);
        } else {
#pragma unroll
            for (int i = 0; i < 64; i += 8) {
                v4u t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = __builtin_nontemporal_load(&in[base + (i + j) * 256]);
#pragma unroll
                for (int j = 0; j < 8; ++j) __builtin_nontemporal_store(t[j], &out[base + (i + j) * 256]);
            }
        }
    }
    if (mode == 0 && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) *sink = 1;
}

// ---------------------------------------------------------------- parts B, C
struct Ctl {
    uint32_t registered;   // workgroups that have taken a slot
    uint32_t error;        // bit 0 spin timeout, bit 1 bad XCD population, bit 2 data mismatch
    uint32_t mismatches;
    uint32_t pad0[13];
    uint32_t xcc_count[16];      // workgroups per XCD
    uint32_t bar[16 * 32];       // one barrier counter per XCD, 128 B apart
    uint32_t flags[16 * 64];     // L2-local barrier: one epoch word per workgroup, 256 B per XCD
    uint32_t t_exchange_ticks[16];
};

template <bool L2SCOPE>
__device__ __forceinline__ uint32_t ctr_add(uint32_t *p)
{
    if (L2SCOPE) return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool L2SCOPE>
__device__ __forceinline__ uint32_t ctr_read(uint32_t *p)
{
    // L2 scope: a returning atomic executed in this XCD's L2 (a workgroup-scope LOAD could be served by L1)
    if (L2SCOPE) return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// XCD-local barrier without atomics: every workgroup publishes its epoch with a PLAIN store (the line stays in this
// XCD's L2), one wave polls all 64 flags with ONE sc1 (L1-bypassing, L2-served) load instruction, lane i reading the
// flag of workgroup i.  Valid only because the 64 workgroups were grouped by HW_REG_XCC_ID (one L2).
__device__ __forceinline__ bool flag_barrier(uint32_t *flags /* 64 words of this XCD */, uint32_t m, uint32_t epoch, uint32_t tid,
                                             uint32_t *err)
{
    if (tid == 0) __hip_atomic_store(&flags[m], epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);  // plain store
    bool ok = true;
    if (tid < 64) {
        auto r = __builtin_amdgcn_make_buffer_rsrc(flags, 0, 256, 0x00020000);
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const uint32_t v = __builtin_amdgcn_raw_buffer_load_b32(r, tid * 4, 0, SC1);
            if (__all((int)(v - epoch) >= 0)) break;
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) {
                __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
    return ok;
}

template <bool L2SCOPE>
__device__ __forceinline__ bool spin_ge(uint32_t *p, uint32_t target, uint32_t *err)
{
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (ctr_read<L2SCOPE>(p) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) {  // 200 ms
            __hip_atomic_fetch_or(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// R rounds of K = 32/R registers.  mailbox (per XCD): [buf 2][dst 64][i K][half 2][256 lanes] x 8 B.
// hbm: 0 = exchange only; 1 = per transform: load the tile (32 x 8 B per thread, column-tile pattern) from `src`,
// exchange, store it to `dst` in the pass-2 pattern.
template <int R, bool L2SCOPE, int hbm, int verify>
__global__ __launch_bounds__(512, 4) void k_xcd_exchange(Ctl *ctl, char *mailbox_all, const char *src, char *dst, uint32_t n_iter,
                                                         uint32_t active_xcds, uint32_t tokens)
{
    constexpr int K = 32 / R;
    __shared__ uint32_t s_info[4];
    const uint32_t tid = threadIdx.x;
    if (tid == 0) {
        const uint32_t x = xcc_id();
        const uint32_t slot = __hip_atomic_fetch_add(&ctl->xcc_count[x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(&ctl->registered, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = spin_ge<false>(&ctl->registered, gridDim.x, &ctl->error);
        const uint32_t pop = __hip_atomic_load(&ctl->xcc_count[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pop != 64) { __hip_atomic_fetch_or(&ctl->error, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); ok = false; }
        s_info[0] = x; s_info[1] = slot; s_info[2] = ok ? 1u : 0u;
    }
    __syncthreads();
    const uint32_t xcc = s_info[0], m = s_info[1];
    if (!s_info[2] || xcc >= active_xcds) return;
    const uint32_t a = m >> 1;  // pair-group
    char *mailbox = mailbox_all + (size_t)xcc * (2u * 64 * K * 2 * 2048);
    auto rmb = __builtin_amdgcn_make_buffer_rsrc(mailbox, 0, 2u * 64 * K * 2 * 2048, 0x00020000);
    uint32_t *bar = &ctl->bar[xcc * 32];
    const uint32_t half = tid >> 8, lane = tid & 255;

    if (hbm == 2) {  // staggered start: XCD x begins x/8 of a 24-us cycle late, so the HBM phases of the XCDs interleave
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - t0 < 300ull * xcc) __builtin_amdgcn_s_sleep(8);
    }
    v2u x[32];
    uint32_t barrier_no = 0;
    uint64_t ticks = 0;
    for (uint32_t it = 0; it < n_iter; ++it) {
        const uint32_t t = it * 8 + xcc;  // transform index
        if (hbm == 3) {
            // HBM token: at most `tokens` XCDs are in their HBM phase (stores of the previous transform + loads of
            // this one) at a time, so the phases of the XCDs interleave instead of marching in step
            if (m == 0 && tid == 0) {
                const uint64_t t0w = __builtin_amdgcn_s_memrealtime();
                for (;;) {
                    if (__hip_atomic_load(&ctl->pad0[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tokens) {
                        if (__hip_atomic_fetch_add(&ctl->pad0[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tokens) break;
                        __hip_atomic_fetch_sub(&ctl->pad0[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    __builtin_amdgcn_s_sleep(8);
                    if (__builtin_amdgcn_s_memrealtime() - t0w > 20000000ull) { __hip_atomic_fetch_or(&ctl->error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
            __syncthreads();
            const bool ok = flag_barrier(&ctl->flags[xcc * 64], m, barrier_no + 1, tid, &ctl->error);
            if (tid == 0) s_info[3] = ok ? 1u : 0u;
            __syncthreads();
            if (!s_info[3]) return;
            ++barrier_no;
            if (it > 0) {  // deferred stores of the previous transform
                auto rout = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)(t - 8) * (8u << 20), 0, 8u << 20, 0x00020000);
                const uint32_t r2 = tid & 15, k1p = tid >> 4;
#pragma unroll
                for (int j = 0; j < 32; ++j) __builtin_amdgcn_raw_buffer_store_b64(x[j], rout, (k1p * 1024 + r2) * 8, m * 128 + j * 262144, NT);
            }
        }
        if (hbm) {
            auto rin = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(src) + (size_t)t * (8u << 20), 0, 8u << 20, 0x00020000);
            const uint32_t c = tid & 15, q = tid >> 4;
#pragma unroll
            for (int j = 0; j < 32; ++j) x[j] = __builtin_amdgcn_raw_buffer_load_b64(rin, (q * 1024 + c) * 8, m * 128 + j * 262144, NT);
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) x[j] = v2u{(it << 20) | (m << 12) | (uint32_t)(j << 4), tid};
        }
        if (hbm == 3) {  // loads landed, stores drained: hand the token back
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const bool ok = flag_barrier(&ctl->flags[xcc * 64], m, barrier_no + 1, tid, &ctl->error);
            if (tid == 0) s_info[3] = ok ? 1u : 0u;
            __syncthreads();
            if (!s_info[3]) return;
            ++barrier_no;
            if (m == 0 && tid == 0) __hip_atomic_fetch_sub(&ctl->pad0[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const uint64_t t0 = __builtin_amdgcn_s_memtime();
        // ---- exchange: phase p (static register index p), partner pair-group b = (p - a) mod 32
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t buf = barrier_no & 1;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int p = r * K + i;
                const uint32_t b = (uint32_t)(p - (int)a) & 31u;
                const uint32_t dstwg = 2 * b + half;  // threads of half h feed workgroup 2b + h
                // slot of the destination: [buf][dst][i][source index = my position in my pair][lane]
                const uint32_t off = ((((buf * 64 + dstwg) * K + i) * 2 + (m & 1)) * 256 + lane) * 8;
                __builtin_amdgcn_raw_buffer_store_b64(x[p], rmb, off, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (L2SCOPE) {
                const bool ok = flag_barrier(&ctl->flags[xcc * 64], m, barrier_no + 1, tid, &ctl->error);
                if (tid == 0) s_info[3] = ok ? 1u : 0u;
            } else if (tid == 0) {
                ctr_add<false>(bar);
                s_info[3] = spin_ge<false>(bar, 64u * (barrier_no + 1), &ctl->error) ? 1u : 0u;
            }
            __syncthreads();
            if (!s_info[3]) return;
            ++barrier_no;
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const int p = r * K + i;
                // my mailbox: both halves of my threads read what the two workgroups of pair-group b sent to me
                const uint32_t off = ((((buf * 64 + m) * K + i) * 2 + half) * 256 + lane) * 8;
                x[p] = __builtin_amdgcn_raw_buffer_load_b64(rmb, off, 0, SC1);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ticks += __builtin_amdgcn_s_memtime() - t0;
        if (verify && !hbm) {
            uint32_t bad = 0;
#pragma unroll
            for (int p = 0; p < 32; ++p) {
                const uint32_t b = (uint32_t)(p - (int)a) & 31u;
                // sender: workgroup 2b + half (its m&1 = half), its threads of half' = (m & 1) wrote lane `lane`
                const uint32_t sm = 2 * b + half;
                const uint32_t want0 = (it << 20) | (sm << 12) | (uint32_t)(p << 4), want1 = ((m & 1) << 8) | lane;
                bad += (x[p].x != want0) + (x[p].y != want1);
            }
            if (bad) { atomicAdd(&ctl->mismatches, bad); __hip_atomic_fetch_or(&ctl->error, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if (hbm && (hbm != 3 || it + 1 == n_iter)) {
            auto rout = __builtin_amdgcn_make_buffer_rsrc(dst + (size_t)t * (8u << 20), 0, 8u << 20, 0x00020000);
            const uint32_t r2 = tid & 15, k1p = tid >> 4;
#pragma unroll
            for (int j = 0; j < 32; ++j) __builtin_amdgcn_raw_buffer_store_b64(x[j], rout, (k1p * 1024 + r2) * 8, m * 128 + j * 262144, NT);
        }
    }
    if (tid == 0 && m == 0) ctl->t_exchange_ticks[xcc] = (uint32_t)(ticks / (n_iter ? n_iter : 1));
    if (!hbm && !verify) {  // keep the registers alive
        uint32_t s = 0;
#pragma unroll
        for (int p = 0; p < 32; ++p) s ^= x[p].x ^ x[p].y;
        if (s == 0xdeadbeefu) ctl->pad0[0] = s;
    }
}

template <int R, bool L2, int hbm>
static void run_exchange(const char *name, Ctl *ctl, char *mailbox, char *a, char *b, uint32_t n_iter, uint32_t active = 8,
                         uint32_t tokens = 8)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f; Ctl h{};
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(ctl, 0, sizeof(Ctl)));
        CK(hipEventRecord(e0));
        if (rep == 0 && !hbm) hipLaunchKernelGGL((k_xcd_exchange<R, L2, 0, 1>), dim3(512), dim3(512), 0, 0, ctl, mailbox, a, b, n_iter, active, tokens);
        else hipLaunchKernelGGL((k_xcd_exchange<R, L2, hbm, 0>), dim3(512), dim3(512), 0, 0, ctl, mailbox, a, b, n_iter, active, tokens);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(&h, ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
        if (h.error) break;
        if (ms < best) best = ms;
    }
    if (h.error) {
        printf("%-62s error word %u (1 timeout, 2 XCD population != 64, 4 data mismatch: %u) population:", name, h.error, h.mismatches);
        for (int i = 0; i < 8; ++i) printf(" %u", h.xcc_count[i]);
        printf("\n");
        return;
    }
    // per XCD: n_iter transforms in `best` ms
    printf("%-62s %8.3f ms / %u transforms per XCD = %7.2f us per transform per XCD (in-kernel exchange span %.2f us)\n", name, best,
           n_iter, best * 1e3 / n_iter, h.t_exchange_ticks[0] / 100.0);
    fflush(stdout);
}

int main()
{
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
    const size_t big = 8ull << 30;
    char *a, *b; CK(hipMalloc(&a, big)); CK(hipMalloc(&b, big));
    CK(hipMemset(a, 1, big)); CK(hipMemset(b, 2, big));
    uint32_t *counter; CK(hipMalloc(&counter, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

    printf("---- part A: HBM streams (nt, 16 B/lane, 256-KiB chunks from one counter) with k active XCDs; GB/s total (per XCD)\n");
    printf("%-8s %18s %18s %18s\n", "k", "read", "write", "copy (r+w)");
    for (uint32_t k : {1u, 2u, 4u, 8u}) {
        double g[3];
        for (int mode = 0; mode < 3; ++mode) {
            const uint64_t bytes = (mode == 2 ? 4ull : 8ull) << 30;  // per direction
            const uint64_t chunks = bytes / 262144;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemset(counter, 0, 256));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(k_xcd_stream, dim3(256 * 8), dim3(256), 0, 0, (const v4u *)a, (v4u *)b, chunks, k, mode, counter, counter + 16);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            g[mode] = (mode == 2 ? 2.0 : 1.0) * bytes / (best * 1e-3) / 1e9;
        }
        printf("%-8u %9.0f (%6.0f) %9.0f (%6.0f) %9.0f (%6.0f)\n", k, g[0], g[0] / k, g[1], g[1] / k, g[2], g[2] / k);
        fflush(stdout);
    }

    Ctl *ctl; CK(hipMalloc(&ctl, sizeof(Ctl)));
    char *mailbox; CK(hipMalloc(&mailbox, 8ull * 2 * 64 * 32 * 2 * 2048)); CK(hipMemset(mailbox, 0, 8ull * 2 * 64 * 32 * 2 * 2048));
    printf("---- part B: in-XCD exchange only (64 workgroups per XCD hold 8 MiB in registers; 8 XCDs run independent groups)\n");
    run_exchange<8, false, 0>("8 rounds x 4 regs, barrier counter: agent-scope atomics", ctl, mailbox, a, b, 64);
    run_exchange<8, true, 0>("8 rounds x 4 regs, barrier: L2-local flags", ctl, mailbox, a, b, 64);
    run_exchange<4, false, 0>("4 rounds x 8 regs, barrier counter: agent-scope atomics", ctl, mailbox, a, b, 64);
    run_exchange<4, true, 0>("4 rounds x 8 regs, barrier: L2-local flags", ctl, mailbox, a, b, 64);
    run_exchange<2, true, 0>("2 rounds x 16 regs (8 MiB of mailbox > L2), L2 flags", ctl, mailbox, a, b, 64);
    run_exchange<16, true, 0>("16 rounds x 2 regs, barrier: L2-local flags", ctl, mailbox, a, b, 64);
    printf("---- part C: HBM tile load -> exchange -> HBM tile store (whole memory skeleton, 128 transforms per XCD)\n");
    run_exchange<4, true, 2>("4 rounds x 8 regs, L2 flags, with HBM, staggered XCD start", ctl, mailbox, a, b, 128);
    run_exchange<8, true, 2>("8 rounds x 4 regs, L2 flags, with HBM, staggered XCD start", ctl, mailbox, a, b, 128);
    run_exchange<4, false, 2>("4 rounds x 8 regs, agent atomics, with HBM, staggered", ctl, mailbox, a, b, 128);
    run_exchange<8, true, 1>("8 rounds x 4 regs, L2 flags, with HBM", ctl, mailbox, a, b, 128);
    run_exchange<4, true, 1>("4 rounds x 8 regs, L2 flags, with HBM", ctl, mailbox, a, b, 128);
    run_exchange<8, false, 1>("8 rounds x 4 regs, agent atomics, with HBM", ctl, mailbox, a, b, 128);
    printf("---- part D: the same skeleton with only k XCDs active (the others exit): is the 36 us HBM contention?\n");
    for (uint32_t k : {1u, 2u, 4u, 6u, 8u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "4 rounds, L2 flags, with HBM, %u active XCDs", k);
        run_exchange<4, true, 1>(nm, ctl, mailbox, a, b, 128, k);
    }
    printf("---- part E: HBM tokens: at most k XCDs in their HBM phase (previous stores + next loads) at a time\n");
    for (uint32_t k : {2u, 3u, 4u, 5u, 6u, 8u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "4 rounds, L2 flags, with HBM, %u tokens", k);
        run_exchange<4, true, 3>(nm, ctl, mailbox, a, b, 128, 8, k);
    }
    printf("budget at 70 %% of the 8 TB/s roofline: 23.4 us per transform per XCD; two-pass pipeline today: 41.6 us\n");
    return 0;
}
