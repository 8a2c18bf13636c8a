// Autoregressive sampler: one persistent kernel of single-wavefront ACTORS (include/aewavenet.h, "Autoregressive
// sampler").  Replaces the per-sample Python loop of WaveNet.forward_test (wavenet.py:367-531).
//
// Why this shape on MI355X.  One output sample needs ~25 MFLOP through 20 strictly sequential layers: nothing for
// the MFMA pipes, everything for latency.  A kernel launch costs ~2.2 us and a grid-wide barrier ~11 us
// (tools/ubench/flag_latency), but a flag hand-off between two wavefronts through L2 costs 0.6-0.9 us.  So the
// generation is ONE launch; all weights (25 MB bf16) stay in the VGPRs of ~1000 resident wavefronts (the chip has
// 128 MB of them), each wavefront owning a 32-channel output slice of one layer; the only per-sample traffic is
// the 16-stream activation rows and the flags.  The critical path per sample is 2 hand-offs per layer (gate, then
// residual) + 4 at the output stack; with several stream-batches in flight the layers pipeline.
//
// MFMA mapping (v_mfma_f32_16x16x32_bf16): A = weight fragment (16 output channels x 32 K), B = activations
// (32 K x 16 streams), D[m][n]: lane holds channels (lane>>4)*4 + {0..3} of stream lane&15.
#include "aew_common.h"

#define SMP_SPIN_DEFAULT (1 << 21)
#define SMP_KD_MAX 8                                   // K tiles of z / skip / post rows (<= 256 channels)
#define SMP_KC_MAX 4                                   // K tiles of the cond row (<= 128 channels)

struct SmpEnv {
    int lane, i, g;                                    // i = stream in the batch, g = K chunk / channel group
    int nb, fstride, spin_max;
    uint32_t* status;
    int slot;
    // Napping.  ~1000 actors polling at once saturate the memory-side path their flags and rows travel on (a row load
    // took 2 us under that load, 0.2 us without).  An actor's items arrive periodically (once per sample per batch), so
    // after publishing one it sleeps through `nap_num`/8 of the running average of its own period before it polls.
    mutable unsigned long long t_pub, period;
    int nap_num;
    uint64_t* prof;                                    // optional phase clock of this actor (aew_sampler_t.prof)
    mutable unsigned long long t_mark, acc_wait, acc_work, acc_pub, items;
    __device__ __forceinline__ void lap(int phase) const {       // 0 after the wait, 1 after the stores, 2 after the flag
        if (!prof) return;
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        const unsigned long long d = now - t_mark;
        t_mark = now;
        if (phase == 0) acc_wait += d;
        else if (phase == 1) acc_work += d;
        else { acc_pub += d; ++items; }
    }
    __device__ __forceinline__ void dump() const {
        if (prof && lane == 0) { prof[0] = acc_wait; prof[1] = acc_work; prof[2] = acc_pub; prof[3] = items; }
    }
};

// Activation rows and flags travel between wavefronts on different CUs / XCDs.  They are read and written with
// relaxed AGENT-scope atomics (global_load / global_store ... sc1: coherent at the device level, never served from a
// stale L2 line, written through), so publishing a row needs no cache maintenance: the producer waits for its stores
// (s_waitcnt vmcnt(0)) and then stores the flag.  The fence form (buffer_wbl2 / buffer_inv, what an acquire / release
// pair costs on a multi-XCD device) walks the L2 and serialises all ~130 actors of an XCD: 6.7 us per hand-off
// measured, against 0.6-0.9 us for the flag itself.
__device__ __forceinline__ unsigned long long smp_ld8(const void* p) {
    typedef const __attribute__((address_space(1))) unsigned long long* gptr;            // global, not flat
    return __hip_atomic_load((gptr)(uintptr_t)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void smp_st8(void* p, unsigned long long v) {
    typedef __attribute__((address_space(1))) unsigned long long* gptr;
    __hip_atomic_store((gptr)(uintptr_t)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
struct SmpU16 { unsigned long long lo, hi; };
__device__ __forceinline__ SmpU16 smp_ld16(const void* p) {
    SmpU16 r;
    r.lo = smp_ld8(p);
    r.hi = smp_ld8(reinterpret_cast<const char*>(p) + 8);
    return r;
}
__device__ __forceinline__ void smp_st16(void* p, const SmpU16& v) {     // one write-through 16-byte store
    // (the s_nop covers the "VALU write of the data registers right after a >8-byte VMEM store" hazard, which the
    // compiler pads for its own stores but cannot see inside inline asm - without it the rows were corrupted)
    const f32x4_t d = __builtin_bit_cast(f32x4_t, v);
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}
__device__ __forceinline__ f32x4_t smp_ld_f4(const void* p) { return __builtin_bit_cast(f32x4_t, smp_ld16(p)); }
__device__ __forceinline__ void smp_st_f4(void* p, const f32x4_t& v) { smp_st16(p, __builtin_bit_cast(SmpU16, v)); }
__device__ __forceinline__ bf16x8_t smp_ld_x(const void* p) { return __builtin_bit_cast(bf16x8_t, smp_ld16(p)); }
__device__ __forceinline__ uint2 smp_ld_u2(const void* p) { return __builtin_bit_cast(uint2, smp_ld8(p)); }
__device__ __forceinline__ void smp_st_u2(void* p, uint2 v) { smp_st8(p, __builtin_bit_cast(unsigned long long, v)); }

__device__ __forceinline__ char* sbuf_at(const aew_sbuf_t& s, int b, int t) {
    return reinterpret_cast<char*>(s.ptr) + (int64_t)b * s.bstride + (int64_t)(t % s.ring) * s.entry;
}

// Wait sets 0 / 1 are polled by lanes 0..31 / 32..63, one flag per lane.  Returns false if the generation was
// aborted (by this actor after spin_max polls, or by another one).
__device__ __forceinline__ bool smp_wait(const aew_actor_t& a, const SmpEnv& e, int t, int b) {
    const int set = e.lane >> 5, j = e.lane & 31;
    const uint32_t* fl = set ? a.wait[1].flags : a.wait[0].flags;
    const int n = set ? a.wait[1].n : a.wait[0].n, lag = set ? a.wait[1].lag : a.wait[0].lag;
    const int tt = t - lag;
    const uint32_t need = (fl && j < n && tt >= 0) ? (uint32_t)(tt * e.nb + b + 1) : 0u;
    const uint32_t* p = fl ? fl + (int64_t)j * e.fstride : nullptr;
    if (e.nap_num > 0 && e.period != 0ull) {
        const unsigned long long wake = e.t_pub + ((e.period * (unsigned long long)e.nap_num) >> 3);
        for (int n = 0; n < 4096 && __builtin_amdgcn_s_memtime() < wake; ++n) __builtin_amdgcn_s_sleep(16);
    }
    if (__any(need != 0u)) {
        bool ok = false;
        for (int spin = 0; spin < e.spin_max; ++spin) {
            const uint32_t v = need ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            if (!__any(v < need)) { ok = true; break; }
            if ((spin & 63) == 63 &&
                __hip_atomic_load(e.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
            __builtin_amdgcn_s_sleep(1);                // (keeping several polls in flight was measured slower: ~1000
        }                                               //  waiting wavefronts then congest the path the flags travel)
        if (!ok) {
            if (e.lane == 0 && atomicCAS(e.status, 0u, 1u) == 0u) {
                e.status[1] = (uint32_t)e.slot; e.status[2] = (uint32_t)t; e.status[3] = (uint32_t)b;
            }
            return false;
        }
    }
    asm volatile("" ::: "memory");                      // the row loads below are coherent (sc1) and stay below
    e.lap(0);
    return true;
}

__device__ __forceinline__ void smp_signal(const aew_actor_t& a, const SmpEnv& e, uint32_t seq) {
    if (e.prof) { asm volatile("s_nop 0" ::: "memory"); e.lap(1); }   // stores issued (work = loads + MFMAs + issue)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the whole wave's (write-through) stores, then the flag
    if (e.lane == 0) __hip_atomic_store(a.flag, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e.nap_num > 0) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        if (e.t_pub != 0ull) {
            const unsigned long long d = now - e.t_pub;
            e.period = e.period == 0ull ? d : (e.period * 3 + d) >> 2;
        }
        e.t_pub = now;
    }
    e.lap(2);
}

template <int KMAX>
struct SmpW {                                          // register-resident weight fragments of one actor
    bf16x8_t w[KMAX][2];
    __device__ __forceinline__ void load(const void* blob, int nk, int lane) {
        const bf16x8_t* p = reinterpret_cast<const bf16x8_t*>(blob);
#pragma unroll
        for (int k = 0; k < KMAX; ++k)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (k < nk) w[k][n] = p[(k * 2 + n) * 64 + lane];
                else w[k][n] = __builtin_bit_cast(bf16x8_t, (s16x8_t){0, 0, 0, 0, 0, 0, 0, 0});
            }
    }
};

// Full-width rows: one coherent 16-byte load per K tile (global_load_dwordx4 sc1; the 8-byte agent-scope atomics of
// smp_ld8 need two), all in flight, then ONE wait that also ties the registers to the MFMAs behind it (the compiler
// does not count loads issued from inline asm).  The data is complete before its flag, so no atomicity is needed here,
// only coherence.
__device__ __forceinline__ void smp_issue16(bf16x8_t& x, const char* p) {
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(x) : "v"(p) : "memory");
}
__device__ __forceinline__ void smp_landed(bf16x8_t (&x)[4]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
}
__device__ __forceinline__ void smp_landed(bf16x8_t (&x)[8]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]));
}
__device__ __forceinline__ void smp_landed(bf16x8_t (&x)[12]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                 "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]));
}
__device__ __forceinline__ void smp_landed(bf16x8_t (&x)[16]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]),
                 "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
}

// acc[n] += W[.. n ..] * x, x = the 16-stream bf16 activations of one entry (`base`) in the buffer's layout: per K tile
// (32 channels) the lane reads the 16 bytes of stream i, channel chunk g (see aew_sbuf_t)
template <int KMAX>
__device__ __forceinline__ void smp_mm(f32x4_t (&acc)[2], const SmpW<KMAX>& W, const char* base, const aew_sbuf_t& sb,
                                       int nk, const SmpEnv& e) {
    const char* lp;
    int ks;
    if (sb.layout == 1) { lp = base + e.i * 64 + e.g * 16; ks = 1024; }
    else if (sb.layout == 2) { lp = base + (e.g >> 1) * 512 + e.i * 32 + (e.g & 1) * 16; ks = 1024; }
    else { lp = base + e.i * sb.pitch + e.g * 16; ks = 64; }
    bf16x8_t x[KMAX];
    if (nk == KMAX) {
        // the usual case (full-width decoder), written without the per-tile `k < nk` tests: with them every K tile is
        // its own basic block and the compiler waits for each block's loads before issuing the next block's
        // (8 trips to memory for a POST1 row, 2 for a RES row; measured in the ISA), instead of one
#pragma unroll
        for (int k = 0; k < KMAX; ++k) smp_issue16(x[k], lp + k * ks);
        smp_landed(x);
        // even / odd K tiles into separate accumulators: four independent MFMA chains of KMAX / 2 instead of two of KMAX
        // (a dependent v_mfma_f32_16x16x32_bf16 issues every ~38 clocks; this is on every hand-off's critical path)
        f32x4_t odd[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < KMAX; k += 2) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.w[k][0], x[k], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.w[k][1], x[k], acc[1], 0, 0, 0);
            odd[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.w[k + 1][0], x[k + 1], odd[0], 0, 0, 0);
            odd[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.w[k + 1][1], x[k + 1], odd[1], 0, 0, 0);
        }
        acc[0] += odd[0];
        acc[1] += odd[1];
        return;
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < nk) x[k] = smp_ld_x(lp + k * ks);
#pragma unroll
    for (int k = 0; k < KMAX; ++k)
        if (k < nk) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.w[k][0], x[k], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.w[k][1], x[k], acc[1], 0, 0, 0);
        }
}

__device__ __forceinline__ uint2 smp_pack4(const f32x4_t& v) {
    const float t[4] = {v[0], v[1], v[2], v[3]};
    return pack4_bf16(t);
}
__device__ __forceinline__ f32x4_t smp_unpack4(uint2 r) {
    return (f32x4_t){bf2f((uint16_t)(r.x & 0xffff)), bf2f((uint16_t)(r.x >> 16)), bf2f((uint16_t)(r.y & 0xffff)),
                     bf2f((uint16_t)(r.y >> 16))};
}

// ---- roles -----------------------------------------------------------------------------------------------------
template <int KR>
__device__ void smp_early(const aew_actor_t& a, const SmpEnv& e, int T) {
    SmpW<KR> W;
    W.load(a.w, a.nk, e.lane);
    const bf16x8_t* w2 = reinterpret_cast<const bf16x8_t*>(a.w2);
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < e.nb; ++b) {
            if (!smp_wait(a, e, t, b)) return;
            f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (t >= a.dil) smp_mm<KR>(acc, W, sbuf_at(a.in0, b, t - a.dil), a.in0, a.nk, e);
            const char* crow = sbuf_at(a.in1, b, t) + e.i * a.in1.pitch;
#pragma unroll
            for (int k = 0; k < SMP_KC_MAX; ++k)
                if (k < a.nk2) {                                     // cond taps: weights streamed (L2), not resident
                    const bf16x8_t x = *reinterpret_cast<const bf16x8_t*>(crow + k * 64 + e.g * 16);
                    acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[(k * 2 + 0) * 64 + e.lane], x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2[(k * 2 + 1) * 64 + e.lane], x, acc[1], 0, 0, 0);
                }
            const float* bp = a.bias + (int64_t)(b * 16 + e.i) * a.bias_pitch + e.g * 4;
            const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(bp), b1 = *reinterpret_cast<const f32x4_t*>(bp + 16);
            char* op = sbuf_at(a.out, b, t) + e.lane * 32;
            smp_st_f4(op, acc[0] + b0);
            smp_st_f4(op + 16, acc[1] + b1);
            smp_signal(a, e, (uint32_t)(t * e.nb + b + 1));
        }
}

template <int KR>
__device__ void smp_late(const aew_actor_t& a, const SmpEnv& e, int T) {
    SmpW<KR> W;
    W.load(a.w, a.nk, e.lane);
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < e.nb; ++b) {
            if (!smp_wait(a, e, t, b)) return;
            const char* pp = sbuf_at(a.in1, b, t) + e.lane * 32;
            f32x4_t acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (a.nk == KR) {                                        // EARLY's partial sums ride along with the row loads
                bf16x8_t p0, p1;
                smp_issue16(p0, pp);
                smp_issue16(p1, pp + 16);
                smp_mm<KR>(acc, W, sbuf_at(a.in0, b, t), a.in0, a.nk, e);     // its wait covers every load in flight
                asm volatile("" : "+v"(p0), "+v"(p1));               // (and nothing reads p0 / p1 before this point)
                acc[0] += __builtin_bit_cast(f32x4_t, p0);
                acc[1] += __builtin_bit_cast(f32x4_t, p1);
            } else {
                acc[0] = smp_ld_f4(pp);
                acc[1] = smp_ld_f4(pp + 16);
                smp_mm<KR>(acc, W, sbuf_at(a.in0, b, t), a.in0, a.nk, e);
            }
            f32x4_t z;
#pragma unroll
            for (int r = 0; r < 4; ++r) z[r] = tanh_f(acc[0][r]) * sigmoid_f(acc[1][r]);
            smp_st_u2(sbuf_at(a.out, b, t) + e.i * a.out.pitch + e.g * 8, smp_pack4(z));
            smp_signal(a, e, (uint32_t)(t * e.nb + b + 1));
        }
}

// RES / SKIP / POST1 / POST2 share the K <= 256 matvec; MODE selects the epilogue
template <int MODE>
__device__ void smp_dense(const aew_actor_t& a, const SmpEnv& e, int T) {
    SmpW<SMP_KD_MAX> W;
    W.load(a.w, a.nk, e.lane);
    f32x4_t bias[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (a.bias) {
        bias[0] = *reinterpret_cast<const f32x4_t*>(a.bias + e.g * 4);
        if (a.nt > 1) bias[1] = *reinterpret_cast<const f32x4_t*>(a.bias + 16 + e.g * 4);
    }
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < e.nb; ++b) {
            if (!smp_wait(a, e, t, b)) return;
            f32x4_t acc[2] = {bias[0], bias[1]};
            typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
            u32x2_t res[2] = {{0u, 0u}, {0u, 0u}};
            bool res_ready = false;
            const char* row = sbuf_at(a.in0, b, t) + e.i * a.in0.pitch;
            if (MODE == AEW_ACT_POST1) {                             // fp32 skip sum -> relu -> bf16 fragments
                bf16x8_t x[SMP_KD_MAX];
                f32x4_t raw[SMP_KD_MAX][2];
                if (a.nk == SMP_KD_MAX) {                            // all 16 loads in flight (see smp_mm)
                    bf16x8_t r16[16];
#pragma unroll
                    for (int k = 0; k < SMP_KD_MAX; ++k) {
                        smp_issue16(r16[2 * k], row + k * 128 + e.g * 32);
                        smp_issue16(r16[2 * k + 1], row + k * 128 + e.g * 32 + 16);
                    }
                    smp_landed(r16);
#pragma unroll
                    for (int k = 0; k < SMP_KD_MAX; ++k) {
                        raw[k][0] = __builtin_bit_cast(f32x4_t, r16[2 * k]);
                        raw[k][1] = __builtin_bit_cast(f32x4_t, r16[2 * k + 1]);
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < SMP_KD_MAX; ++k)
                        if (k < a.nk) {
                            raw[k][0] = smp_ld_f4(row + k * 128 + e.g * 32);
                            raw[k][1] = smp_ld_f4(row + k * 128 + e.g * 32 + 16);
                        }
                }
#pragma unroll
                for (int k = 0; k < SMP_KD_MAX; ++k)
                    if (k < a.nk) {
                        const f32x4_t lo = raw[k][0];
                        const f32x4_t hi = raw[k][1];
                        uint4 u;
                        u.x = pack2_bf16(fmaxf(lo[0], 0.f), fmaxf(lo[1], 0.f));
                        u.y = pack2_bf16(fmaxf(lo[2], 0.f), fmaxf(lo[3], 0.f));
                        u.z = pack2_bf16(fmaxf(hi[0], 0.f), fmaxf(hi[1], 0.f));
                        u.w = pack2_bf16(fmaxf(hi[2], 0.f), fmaxf(hi[3], 0.f));
                        x[k] = __builtin_bit_cast(bf16x8_t, u);
                    }
#pragma unroll
                for (int k = 0; k < SMP_KD_MAX; ++k)
                    if (k < a.nk) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.w[k][0], x[k], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W.w[k][1], x[k], acc[1], 0, 0, 0);
                    }
            } else if (MODE == AEW_ACT_RES && a.nk == SMP_KD_MAX) {
                // the residual taps h_l(t) ride along with the z loads (issued first, tied after the row's wait) instead
                // of costing a second trip to memory after the MFMAs
                const char* rp = sbuf_at(a.in1, b, t) + e.i * a.in1.pitch + e.g * 8;
                asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(res[0]) : "v"(rp) : "memory");
                if (a.nt > 1) asm volatile("global_load_dwordx2 %0, %1, off sc1" : "=&v"(res[1]) : "v"(rp + 32) : "memory");
                smp_mm<SMP_KD_MAX>(acc, W, sbuf_at(a.in0, b, t), a.in0, a.nk, e);
                asm volatile("" : "+v"(res[0]), "+v"(res[1]));
                res_ready = true;
            } else {
                smp_mm<SMP_KD_MAX>(acc, W, sbuf_at(a.in0, b, t), a.in0, a.nk, e);
            }
            char* orow = sbuf_at(a.out, b, t) + e.i * a.out.pitch;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (n >= a.nt) break;
                if (MODE == AEW_ACT_RES) {                           // h_{l+1} = h_l + W z            (wavenet.py:108-110)
                    const uint2 r = res_ready ? __builtin_bit_cast(uint2, res[n])
                                              : smp_ld_u2(sbuf_at(a.in1, b, t) + e.i * a.in1.pitch + n * 32 + e.g * 8);
                    smp_st_u2(orow + n * 32 + e.g * 8, smp_pack4(acc[n] + smp_unpack4(r)));
                } else if (MODE == AEW_ACT_SKIP) {                   // running skip sum, fp32       (wavenet.py:458)
                    f32x4_t prev = {0.f, 0.f, 0.f, 0.f};
                    if (a.in1.ptr) prev = smp_ld_f4(sbuf_at(a.in1, b, t) + e.i * a.in1.pitch + n * 64 + e.g * 16);
                    smp_st_f4(orow + n * 64 + e.g * 16, acc[n] + prev);
                } else if (MODE == AEW_ACT_POST1) {                  // relu(post1(relu(skip)))      (wavenet.py:461-462)
                    f32x4_t v = acc[n];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
                    smp_st_u2(orow + n * 32 + e.g * 8, smp_pack4(v));
                } else {                                             // logits
                    smp_st_f4(orow + n * 64 + e.g * 16, acc[n]);
                    if (a.out2.ptr)
                        *reinterpret_cast<f32x4_t*>(sbuf_at(a.out2, b, t) + e.i * a.out2.pitch + n * 64 + e.g * 16) = acc[n];
                }
            }
            smp_signal(a, e, (uint32_t)(t * e.nb + b + 1));
        }
}

// SAMPLE actor `index` serves streams [4*index, 4*index+4) of every batch: 16 lanes per stream, Q/16 logits per
// lane.  Draw = first k with cumsum(exp(l - max))[k] > u * total  (inverse CDF; u from the counter RNG), which
// replaces softmax + torch.multinomial(probs, 1) (wavenet.py:463-464).
__device__ void smp_sample(const aew_actor_t& a, const SmpEnv& e, const aew_sampler_t& s) {
    const int T = s.n_steps, sub = e.lane >> 4, l16 = e.lane & 15;
    const int per = a.n_quant >> 4;                                  // <= 16
    const int chunks = a.row_bytes >> 4;                             // 16-byte chunks of an h_0 row
    auto feed = [&](int b, int t, int value) {                       // publish position t: wav_out + h_0(t)
        const int stream = b * 16 + a.index * 4 + sub;
        if (l16 == 0) s.wav_out[(int64_t)stream * T + t] = value;
        const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(a.w) + (int64_t)value * a.row_bytes);
        char* dst = sbuf_at(a.out, b, t) + (a.index * 4 + sub) * 64;           // layout 1: [group][stream][32 ch]
        for (int c = l16; c < chunks; c += 16)
            smp_st16(dst + (c >> 2) * 1024 + (c & 3) * 16, __builtin_bit_cast(SmpU16, src[c]));
    };
    for (int b = 0; b < e.nb; ++b) {                                 // position 0 is given
        const int stream = b * 16 + a.index * 4 + sub;
        int v = s.forced[(int64_t)stream * T];
        v = v < 0 ? 0 : (v >= a.n_quant ? a.n_quant - 1 : v);
        feed(b, 0, v);
    }
    smp_signal(a, e, (uint32_t)e.nb);                                // h_0(0, b) ready for every b
    for (int t = 0; t < T; ++t)
        for (int b = 0; b < e.nb; ++b) {
            if (!smp_wait(a, e, t, b)) return;
            if (t + 1 < T) {
                const int stream = b * 16 + a.index * 4 + sub;
                int value = s.forced[(int64_t)stream * T + t + 1];
                const float* lp = reinterpret_cast<const float*>(sbuf_at(a.in0, b, t) + (a.index * 4 + sub) * a.in0.pitch) + l16 * per;
                float v[16];
                float mx = -3.0e38f;
                if (per == 16) {                                     // Q = 256: four 16-byte loads per lane
                    bf16x8_t r4[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) smp_issue16(r4[q], reinterpret_cast<const char*>(lp) + q * 16);
                    smp_landed(r4);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4_t f = __builtin_bit_cast(f32x4_t, r4[q]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) { v[q * 4 + r] = f[r]; mx = fmaxf(mx, f[r]); }
                    }
                }
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (per != 16 && k < per) {
                        if (!(k & 1)) {                              // coherent 8-byte loads (per is even: Q % 32 == 0) or
                            if (k + 1 < per) {                       // a single trailing word
                                const unsigned long long w = smp_ld8(lp + k);
                                v[k] = __uint_as_float((unsigned)w);
                                v[k + 1] = __uint_as_float((unsigned)(w >> 32));
                            } else {
                                v[k] = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(lp + k), __ATOMIC_RELAXED,
                                                                         __HIP_MEMORY_SCOPE_AGENT));
                            }
                        }
                        mx = fmaxf(mx, v[k]);
                    }
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
                float part = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < per) { v[k] = __expf(v[k] - mx); part += v[k]; }
                float incl = part;                                   // inclusive scan over the 16 lanes of the stream
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    const float up = __shfl_up(incl, o, 16);
                    if (l16 >= o) incl += up;
                }
                const float total = __shfl(incl, 15, 16);
                const float u = (float)aew_jitter_u(s.seed, 0x53414d50ull, stream, t + 1);
                const float target = u * total;
                // owner = first lane whose inclusive sum exceeds target (last lane if rounding leaves none)
                const bool mine = incl > target;
                const unsigned long long m = __ballot(mine) >> (e.lane & 48);
                const int owner = (m & 0xffffull) ? __ffsll((long long)(m & 0xffffull)) - 1 : 15;
                int pick = per - 1;
                float run = incl - part;
                bool found = false;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (k < per) {
                        run += v[k];
                        if (!found && run > target) { pick = k; found = true; }
                    }
                const int drawn = __shfl(l16 * per + pick, owner, 16);
                if (value < 0) value = drawn;
                value = value >= a.n_quant ? a.n_quant - 1 : value;
                feed(b, t + 1, value);
            }
            smp_signal(a, e, (uint32_t)((t + 1) * e.nb + b + 1));
        }
}

// two actors per SIMD must fit (a DEEP decoder has 1684): at most 256 registers per wavefront
template <int KR>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_sampler(const aew_sampler_t s) {
    const int slot = blockIdx.x;
    const aew_actor_t a = s.actors[slot];               // by value: wave-uniform, lives in SGPRs
    const int role = __builtin_amdgcn_readfirstlane(a.role);
    if (role < 0) return;
    SmpEnv e;
    e.lane = threadIdx.x; e.i = e.lane & 15; e.g = e.lane >> 4;
    e.nb = s.n_batches; e.fstride = s.flag_stride; e.spin_max = s.spin_max > 0 ? s.spin_max : SMP_SPIN_DEFAULT;
    e.status = s.status; e.slot = slot;
    e.t_pub = e.period = 0ull;
    e.nap_num = s.nap_eighths < 0 ? 0 : (s.nap_eighths > 7 ? 7 : s.nap_eighths);
    e.prof = s.prof ? s.prof + (int64_t)slot * 4 : nullptr;
    e.acc_wait = e.acc_work = e.acc_pub = e.items = 0;
    e.t_mark = s.prof ? __builtin_amdgcn_s_memtime() : 0ull;
    switch (role) {
        case AEW_ACT_EARLY: smp_early<KR>(a, e, s.n_steps); break;
        case AEW_ACT_LATE: smp_late<KR>(a, e, s.n_steps); break;
        case AEW_ACT_RES: smp_dense<AEW_ACT_RES>(a, e, s.n_steps); break;
        case AEW_ACT_SKIP: smp_dense<AEW_ACT_SKIP>(a, e, s.n_steps); break;
        case AEW_ACT_POST1: smp_dense<AEW_ACT_POST1>(a, e, s.n_steps); break;
        case AEW_ACT_POST2: smp_dense<AEW_ACT_POST2>(a, e, s.n_steps); break;
        case AEW_ACT_SAMPLE: smp_sample(a, e, s); break;
        default: break;
    }
    e.dump();
}

static int launch_sampler(const aew_sampler_t& s, hipStream_t st) {
    if (!s.actors || !s.flags || !s.status || !s.forced || !s.wav_out) return AEW_E_ARG;
    if (s.n_slots < 1 || s.n_batches < 1 || s.n_steps < 1 || s.flag_stride < 1) return AEW_E_ARG;
    if (s.kr_max != 12 && s.kr_max != 16) return AEW_E_ARG;
    if ((int64_t)s.n_steps * s.n_batches >= (1ll << 31) - 2) return AEW_E_ARG;       // sequence numbers are 32-bit
    const void* fn = s.kr_max == 12 ? reinterpret_cast<const void*>(k_sampler<12>) : reinterpret_cast<const void*>(k_sampler<16>);
    // every actor must be resident at once: they wait on each other
    int dev = 0, per_cu = 0, cus = 0;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess) return (int)err;
    err = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (err != hipSuccess) return (int)err;
    err = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64, 0);
    if (err != hipSuccess) return (int)err;
    if ((int64_t)per_cu * cus < s.n_slots) return AEW_E_UNSUP;
    err = hipMemsetAsync(s.flags, 0, (size_t)s.n_slots * s.flag_stride * sizeof(uint32_t), st);
    if (err != hipSuccess) return (int)err;
    err = hipMemsetAsync(s.status, 0, 4 * sizeof(uint32_t), st);
    if (err != hipSuccess) return (int)err;
    if (s.kr_max == 12) hipLaunchKernelGGL(k_sampler<12>, dim3(s.n_slots), dim3(64), 0, st, s);
    else hipLaunchKernelGGL(k_sampler<16>, dim3(s.n_slots), dim3(64), 0, st, s);
    return (int)hipGetLastError();
}
