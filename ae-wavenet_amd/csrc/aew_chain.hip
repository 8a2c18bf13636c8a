// aew_chain.hip — a run of DEPENDENT bf16 NT GEMMs as ONE launch (gfx950): AEW_OP_NT_CHAIN, aew_nt_chain_t.
//
// wavenet.py:354-357 is a loop over layers, each of them two GEMMs (:100-102 gated pair, :108-109 residual 1x1) that
// read what the previous one wrote; its backward is the same loop in reverse (dz, dx per layer).  As stand-alone
// launches each of those 80 GEMMs pays the fill and the drain of a dependent launch - ~3 us until the first workgroups
// run, a cold first tile per CU, a last tile per CU that runs alone: ~10 us beyond its tiles (profiles/r04_notes.md 13).
// Here the tiles of ALL stages of a run form one grid, in stage order, and the dependency between stages is kept per
// ROW TILE instead of per launch:
//
//   * a tile of stage s reads rows [m0 + d_lo, m_last + d_hi] of what an earlier stage p wrote (aew_chain_dep_t, derived
//     on the host from the segment / view records: the dilated tap of wavenet.py:100 is d_hi = dilation);
//   * stage p keeps one counter per (batch element, row tile); each of its N tiles adds 1 after its out0 rows - stored
//     WRITE-THROUGH (sc1, store16_wt) - have drained (`s_waitcnt vmcnt(0)` in every wave, workgroup barrier, one
//     device-scope atomic);
//   * a consumer tile polls the counters it needs with one wave (one lane per counter, relaxed device-scope loads,
//     s_sleep between polls, bounded), a workgroup barrier, and the kernel body - which, as a stage of a chain, reads its
//     activation pieces (LDS-DMA) and epilogue operands DEVICE-SCOPE (sc1: never from the CU's L1), the form that stands
//     in for an agent-scope acquire when the producer stored sc1 (the acquire itself: +0.05 ms per step, flags & 4).
//
// This is the guide's publish / consume recipe (cdna_hip_programming.md Guideline 16, form R1): nothing depends on
// which XCD a workgroup lands on.  What the bounded spin relies on is that workgroups are dispatched in index order:
// every tile a resident tile waits for has a LOWER index (stage order), so it is resident or done - waits cannot
// deadlock, and since a stage is 600-900 tiles and 512 are resident, a tile's producers have normally finished long
// before it starts (the poll is one round trip).  A wait that gives up sets counters[n_counters] and the tile runs on.
//
// Kernel bodies: nt_tile<EPI, .., 4, 1, 256, 3> and win_tile<EPI, 4, 1 | 4> - the stand-alone launches' own code,
// same tile order inside a stage (the N tiles of a row tile consecutive, block L of a stage on XCD L % 8 since stages
// start at multiples of 8), same summation order: bit-identical to the serial plan.
#pragma once

#define CHAIN_BM 256
#define CHAIN_THREADS 512
#define CHAIN_LDS_BYTES (NtCfg<4, 1, 256>::LDS_BYTES > WinCfg<4, 4>::LDS_BYTES ? NtCfg<4, 1, 256>::LDS_BYTES : WinCfg<4, 4>::LDS_BYTES)
static_assert(WinCfg<4, 1>::LDS_BYTES <= CHAIN_LDS_BYTES && 2 * CHAIN_LDS_BYTES <= 160 * 1024, "two chain blocks per CU");

// producer row tiles [t_lo, t_hi] a consumer tile with rows [m0, m_last] waits for (empty: t_lo > t_hi)
__host__ __device__ __forceinline__ void chain_dep_tiles(const aew_chain_dep_t& d, int m0, int m_last, int& t_lo, int& t_hi) {
    int a = m0 + d.d_lo, b = m_last + d.d_hi;
    a = a > d.c_lo ? a : d.c_lo;
    b = b < d.c_hi ? b : d.c_hi;
    if (a > b) { t_lo = 1; t_hi = 0; return; }
    t_lo = a / d.bm;
    t_hi = b / d.bm;
}

template <int SET>
__global__ __launch_bounds__(CHAIN_THREADS, 4) void k_nt_chain(const aew_nt_stage_t* __restrict__ stages,
                                                              const uint16_t* __restrict__ block_stage,
                                                              unsigned* counters, int n_counters, int spin_max, int flags,
                                                              unsigned* sticky) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s = __builtin_amdgcn_readfirstlane((int)block_stage[blockIdx.x >> 3]);
    const aew_nt_stage_t& S = stages[s];
    const int L = (int)blockIdx.x - S.first_block;
    const int n_mt = S.n_mt, n_nt = S.n_nt;
    const int rt = ((L >> 3) / n_nt) * 8 + (L & 7);           // the bodies' own tile order
    if (rt >= n_mt * S.g.batch) return;                       // padding block of the stage's last group of 8
    const int b = rt / n_mt;
    const int m0 = (rt - b * n_mt) * CHAIN_BM;
    const int m_last = (m0 + CHAIN_BM < S.g.M ? m0 + CHAIN_BM : S.g.M) - 1;
    const int tid = threadIdx.x;
    const int n_deps = S.n_deps;
    if (n_deps > 0) {
        if (tid < 64) {                                       // wave 0: one lane per counter
            int idx = -1;
            unsigned need = 0;
            int base = 0;
            for (int d = 0; d < n_deps; ++d) {                // (wave-uniform)
                const aew_chain_dep_t D = S.dep[d];
                int lo, hi;
                chain_dep_tiles(D, m0, m_last, lo, hi);
                const int n = hi >= lo ? hi - lo + 1 : 0;
                const int k = tid - base;
                if (k >= 0 && k < n) { idx = D.cnt_base + b * D.n_mt + lo + k; need = (unsigned)D.need; }
                base += n;
            }
            if (base > 0) {
                bool ok = idx < 0;
                int spins = 0;
                for (;;) {
                    if (!ok) ok = __hip_atomic_load(counters + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= need;
                    if (__all(ok)) break;
                    if (++spins >= spin_max) {                // a producer that never arrives: report, run on
                        if (tid == 0) {
                            __hip_atomic_store(counters + n_counters, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            // ... and in the word no launch clears: the optimizer's guard (aew_adam_t.guard) - the step does
                            // not reach the parameters, and the host finds it whenever it looks next
                            if (sticky) __hip_atomic_fetch_max(sticky, (unsigned)(s + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        break;
                    }
                    // (once any wait of the launch has given up the others do not sit out their own limit)
                    if ((spins & 255) == 0 && __hip_atomic_load(counters + n_counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
                    __builtin_amdgcn_s_sleep(4);
                }
                if (spins > 0 && tid == 0) {                  // statistics of the launch: tiles that had to wait, longest wait (polls)
                    __hip_atomic_fetch_add(counters + n_counters + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_fetch_max(counters + n_counters + 2, (unsigned)spins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                // No acquire fence: everything a stage reads that an earlier stage of this launch may have written - the
                // activation pieces of the K loop and the epilogue's aux operands - is loaded device-scope (sc1: never from
                // this CU's L1), and the producer stored it write-through and drained before raising the counter
                // (Guideline 16: sc1 loads stand in for the acquire when the producer stored sc1).  flags & 4 adds the
                // fence back (A/B: 0.05 ms per step over both directions).
                if (flags & 4) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
        }
        __syncthreads();
    }
    const int kind = S.kind, epi = S.g.epi;
    if (SET == 0) {
        if (epi == AEW_EPI_GATED) {
            if (kind == 1) win_tile<AEW_EPI_GATED, 4, 1, true>(S.g, smem, L);
            else if (kind == 2) win_tile<AEW_EPI_GATED, 4, 4, true>(S.g, smem, L);
            else nt_tile<AEW_EPI_GATED, false, 4, 1, 256, NT_STAGES, true>(S.g, smem, L);
        } else {
            nt_tile<AEW_EPI_STORE, false, 4, 1, 256, NT_STAGES, true>(S.g, smem, L);
        }
    } else {
        if (epi == AEW_EPI_DFG) {
            nt_tile<AEW_EPI_DFG, false, 4, 1, 256, NT_STAGES, true>(S.g, smem, L);
        } else {
            if (kind == 1) win_tile<AEW_EPI_STORE, 4, 1, true>(S.g, smem, L);
            else if (kind == 2) win_tile<AEW_EPI_STORE, 4, 4, true>(S.g, smem, L);
            else nt_tile<AEW_EPI_STORE, false, 4, 1, 256, NT_STAGES, true>(S.g, smem, L);
        }
    }
    if (S.publish) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
        __syncthreads();                                      // ... all of them have
        if (tid == 0) __hip_atomic_fetch_add(counters + S.cnt_base + rt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- host side ----------------------------------------------------------------------------------------------
#include <vector>

static int win_dwp(const aew_gemm_nt_t& g);
static int check_seg(const aew_seg_t& s, int esize, int ktile);
static int launch_zero(const aew_zero_t& z, hipStream_t st);

struct ChainRef {                                            // one operand / result of a stage as rows of a buffer
    const char* ptr; int64_t bs; int pitch, step, off, lo, hi, es;
};
struct ChainHull { const char* a; const char* b; };          // [a, b) bytes; a == nullptr: nothing accessed

static ChainHull chain_hull(const ChainRef& r, int M, int batch) {
    const int64_t r0 = r.off, r1 = (int64_t)(M - 1) * r.step + r.off;
    int64_t rmin = r0 < r1 ? r0 : r1, rmax = r0 < r1 ? r1 : r0;
    if (rmin < r.lo) rmin = r.lo;
    if (rmax > (int64_t)r.hi - 1) rmax = (int64_t)r.hi - 1;
    if (!r.ptr || rmin > rmax) return {nullptr, nullptr};
    return {r.ptr + rmin * r.pitch * r.es, r.ptr + ((int64_t)(batch - 1) * r.bs + (rmax + 1) * r.pitch) * r.es};
}
static bool chain_overlap(const ChainHull& x, const ChainHull& y) { return x.a && y.a && x.a < y.b && y.a < x.b; }

static ChainRef chain_ref(const aew_seg_t& s) {
    return {reinterpret_cast<const char*>(s.ptr), s.batch_stride, s.row_pitch, s.row_step, s.row_off, s.row_lo, s.row_hi, 2};
}
static ChainRef chain_ref(const aew_view_t& v) {
    return {reinterpret_cast<const char*>(v.ptr), v.batch_stride, v.row_pitch, v.row_step, v.row_off, v.row_lo, v.row_hi,
            v.dtype == AEW_BF16 ? 2 : 4};
}
static void chain_inputs(const aew_gemm_nt_t& g, std::vector<ChainRef>& v) {
    for (int s = 0; s < g.n_segs; ++s) v.push_back(chain_ref(g.seg[s]));
    const bool a0 = g.epi == AEW_EPI_DFG || (g.epi == AEW_EPI_STORE && (g.flags & AEW_EF_ADD_AUX0));
    const bool a1 = g.epi == AEW_EPI_DFG || (g.epi == AEW_EPI_STORE && (g.flags & (AEW_EF_MUL_POS1 | AEW_EF_OUT1_POS1)));
    if (a0 && g.aux0.ptr) v.push_back(chain_ref(g.aux0));
    if (a1 && g.aux1.ptr) v.push_back(chain_ref(g.aux1));
}
static void chain_side_outputs(const aew_gemm_nt_t& g, std::vector<ChainRef>& v) {      // everything but out0
    const bool o1 = g.epi == AEW_EPI_GATED || (g.epi == AEW_EPI_STORE && (g.flags & (AEW_EF_OUT1_PRE | AEW_EF_OUT1_POS1)));
    const bool o2 = g.epi == AEW_EPI_GATED;
    if (o1 && g.out1.ptr) v.push_back(chain_ref(g.out1));
    if (o2 && g.out2.ptr) v.push_back(chain_ref(g.out2));
}

// a descriptor the chain's bodies execute exactly as launch_gemm_nt's default shape would
static int chain_desc_ok(const aew_gemm_nt_t& g, int force) {
    if (g.dtype != AEW_BF16 || g.impl != 0 || g.W2 || !g.W || g.M <= 0 || g.batch <= 0) return AEW_E_UNSUP;
    if (g.n_segs < 1 || g.n_segs > AEW_MAX_SEGS) return AEW_E_ARG;
    if (g.epi != AEW_EPI_STORE && g.epi != AEW_EPI_GATED && g.epi != AEW_EPI_DFG) return AEW_E_UNSUP;
    int ksum = 0;
    for (int s = 0; s < g.n_segs; ++s) {
        const int rc = check_seg(g.seg[s], 2, 64);
        if (rc) return rc;
        if (g.seg[s].k_len * 2 > AEW_ZERO_SPAN) return AEW_E_UNSUP;
        ksum += g.seg[s].k_len;
    }
    if (ksum != g.K_total || g.N_pad % NT_BN || g.N > g.N_pad || (g.N & 7)) return AEW_E_ARG;
    if ((g.epi == AEW_EPI_STORE || g.epi == AEW_EPI_DFG) &&
        ((g.aux0.ptr && g.aux0.dtype != AEW_BF16) || (g.aux1.ptr && g.aux1.dtype != AEW_BF16)))
        return AEW_E_UNSUP;
    if (g.epi == AEW_EPI_STORE && (g.flags & (AEW_EF_OUT2_COPY | AEW_EF_COUNT_ZERO))) return AEW_E_UNSUP;
    if (!g.out0.ptr) return AEW_E_ARG;
    // a chained stage loads its epilogue operands through a raw buffer resource with a 32-bit byte offset from the view's
    // base (ld16_sc1): a view whose extent reaches 2 GiB would read zeros (out of range) or a wrapped address
    for (const aew_view_t* v : {&g.aux0, &g.aux1}) {
        if (!v->ptr) continue;
        const int64_t rows = v->row_hi > 0 ? v->row_hi : 0;
        const int64_t extent = ((int64_t)(g.batch - 1) * v->batch_stride + rows * v->row_pitch) * (v->dtype == AEW_BF16 ? 2 : 4);
        if (extent < 0 || extent > 0x7fffffffLL) return AEW_E_UNSUP;
    }
    const aew_tuning_t& T = AEW_T();
    if (T.nt_wave_rows != 64 || T.nt_mem128 || T.nt_deep) return AEW_E_UNSUP;          // an A/B shape is selected
    const int tiles256 = ((g.M + NT_BM - 1) / NT_BM) * g.batch * (g.N_pad / NT_BN);
    if (!force && T.nt_small_tiles > 0 && tiles256 <= T.nt_small_tiles) return AEW_E_UNSUP;   // small launches: 64-row tiles
    return 0;
}

extern "C" int aew_nt_chain_dep_tiles(const aew_nt_stage_t* stage, int dep, int m0, int* t_lo, int* t_hi) {
    if (!stage || dep < 0 || dep >= stage->n_deps || !t_lo || !t_hi || m0 < 0 || m0 >= stage->g.M) return AEW_E_ARG;
    const int m_last = (m0 + CHAIN_BM < stage->g.M ? m0 + CHAIN_BM : stage->g.M) - 1;
    chain_dep_tiles(stage->dep[dep], m0, m_last, *t_lo, *t_hi);
    return 0;
}

extern "C" int aew_nt_chain_build(const aew_gemm_nt_t* descs, int n, aew_nt_stage_t* out, uint16_t* block_stage,
                                  int cap_blocks, int* n_blocks_out, int* n_counters_out, int* set_out, int force) {
    if (!descs || !out || !block_stage || n < 1 || n > 65535 || !n_blocks_out || !n_counters_out || !set_out) return AEW_E_ARG;
    bool has_gated = false, has_dfg = false, store_win = false;
    int blocks = 0, counters = 0;
    struct FullDep { int p; aew_chain_dep_t d; };
    std::vector<std::vector<FullDep>> full(n);
    for (int s = 0; s < n; ++s) {
        const aew_gemm_nt_t& g = descs[s];
        const int rc = chain_desc_ok(g, force);
        if (rc) return rc;
        aew_nt_stage_t& S = out[s];
        S = aew_nt_stage_t{};
        S.g = g;
        const int dwp = win_dwp(g);
        S.kind = dwp == 0 ? 0 : (dwp == 1 ? 1 : 2);
        has_gated = has_gated || g.epi == AEW_EPI_GATED;
        has_dfg = has_dfg || g.epi == AEW_EPI_DFG;
        store_win = store_win || (g.epi == AEW_EPI_STORE && S.kind != 0);
        S.n_mt = (g.M + CHAIN_BM - 1) / CHAIN_BM;
        S.n_nt = g.N_pad / NT_BN;
        S.first_block = blocks;
        S.n_blocks = ((S.n_mt * g.batch + 7) / 8) * 8 * S.n_nt;
        S.cnt_base = counters;
        blocks += S.n_blocks;
        counters += S.n_mt * g.batch;
        if (g.batch != descs[0].batch) return AEW_E_UNSUP;                    // counters are indexed by batch element
        // ---- what this stage reads and writes against every earlier stage
        std::vector<ChainRef> ins, outs_s;
        chain_inputs(g, ins);
        const ChainRef o0 = chain_ref(g.out0);
        chain_side_outputs(g, outs_s);
        for (int p = 0; p < s; ++p) {
            const aew_gemm_nt_t& gp = descs[p];
            const ChainRef po = chain_ref(gp.out0);
            const ChainHull pho = chain_hull(po, gp.M, gp.batch);
            std::vector<ChainRef> pside, pins;
            chain_side_outputs(gp, pside);
            chain_inputs(gp, pins);
            // write-after-read / write-after-write inside the run: refused (every stage of the stack has its own buffers)
            std::vector<ChainRef> my_outs = outs_s;
            my_outs.push_back(o0);
            for (const ChainRef& mo : my_outs) {
                const ChainHull h = chain_hull(mo, g.M, g.batch);
                if (chain_overlap(h, pho)) return AEW_E_UNSUP;
                for (const ChainRef& q : pside) if (chain_overlap(h, chain_hull(q, gp.M, gp.batch))) return AEW_E_UNSUP;
                for (const ChainRef& q : pins) if (chain_overlap(h, chain_hull(q, gp.M, gp.batch))) return AEW_E_UNSUP;
            }
            bool have = false;
            aew_chain_dep_t D = {};
            for (const ChainRef& in : ins) {
                const ChainHull h = chain_hull(in, g.M, g.batch);
                for (const ChainRef& q : pside)                                // only out0 is published write-through
                    if (chain_overlap(h, chain_hull(q, gp.M, gp.batch))) return AEW_E_UNSUP;
                if (!chain_overlap(h, pho)) continue;
                if (in.ptr != po.ptr || in.bs != po.bs || in.pitch != po.pitch || in.es != po.es || in.step != 1 || po.step != 1)
                    return AEW_E_UNSUP;
                const int delta = in.off - po.off;                             // consumer row m reads producer row m + delta
                const int r_lo = in.lo > po.lo ? in.lo : po.lo, r_hi = in.hi < po.hi ? in.hi : po.hi;   // buffer rows both sides know
                int c_lo = r_lo - po.off, c_hi = r_hi - 1 - po.off;
                if (c_lo < 0) c_lo = 0;
                if (c_hi > gp.M - 1) c_hi = gp.M - 1;
                if (c_lo > c_hi) continue;
                if (!have) { D.d_lo = D.d_hi = delta; D.c_lo = c_lo; D.c_hi = c_hi; have = true; }
                else {
                    if (delta < D.d_lo) D.d_lo = delta;
                    if (delta > D.d_hi) D.d_hi = delta;
                    if (c_lo < D.c_lo) D.c_lo = c_lo;
                    if (c_hi > D.c_hi) D.c_hi = c_hi;
                }
            }
            if (have) {
                D.cnt_base = out[p].cnt_base; D.n_mt = out[p].n_mt; D.need = out[p].n_nt; D.bm = CHAIN_BM;
                full[s].push_back({p, D});
            }
        }
    }
    if ((has_gated && has_dfg) || (has_gated && store_win)) return AEW_E_UNSUP;   // one body set per launch
    if (blocks > cap_blocks) return AEW_E_ARG;
    // ---- drop dependencies that another one implies: s waits for q, every tile of q it waits for has itself waited
    // for the tiles of p that s needs (dx of a layer reads dx of the layer above through dz's rows)
    auto need = [&](const aew_chain_dep_t& d, int M, int mt, int& lo, int& hi) {
        const int m0 = mt * CHAIN_BM, m_last = (m0 + CHAIN_BM < M ? m0 + CHAIN_BM : M) - 1;
        chain_dep_tiles(d, m0, m_last, lo, hi);
    };
    for (int s = 0; s < n; ++s) {
        aew_nt_stage_t& S = out[s];
        const std::vector<FullDep>& F = full[s];
        for (size_t i = 0; i < F.size(); ++i) {
            const int p = F[i].p;
            bool implied = false;
            for (size_t j = 0; j < F.size() && !implied; ++j) {
                const int q = F[j].p;
                if (q <= p) continue;
                const aew_chain_dep_t* qp = nullptr;                           // q's own dependency on p
                for (const FullDep& fd : full[q]) if (fd.p == p) qp = &fd.d;
                if (!qp) continue;
                bool all = true;
                for (int mt = 0; mt < S.n_mt && all; ++mt) {
                    int lo, hi, qlo, qhi;
                    need(F[i].d, S.g.M, mt, lo, hi);
                    need(F[j].d, S.g.M, mt, qlo, qhi);
                    for (int t = lo; t <= hi && all; ++t) {
                        bool cov = false;
                        for (int u = qlo; u <= qhi && !cov; ++u) {
                            int plo, phi;
                            need(*qp, descs[q].M, u, plo, phi);
                            cov = t >= plo && t <= phi;
                        }
                        all = cov;
                    }
                }
                implied = all;
            }
            if (implied) continue;
            if (S.n_deps >= AEW_CHAIN_MAXDEP) return AEW_E_UNSUP;
            S.dep[S.n_deps++] = F[i].d;
            out[p].publish = 1;
        }
        // a pruned dependency still needs its producer's data in memory: it is, because the stage that implies it
        // waited for that producer's counter, which is only raised after the write-through stores have drained
        for (int mt = 0; mt < S.n_mt; ++mt) {                                  // one lane of wave 0 per counter
            int cnt = 0;
            for (int d = 0; d < S.n_deps; ++d) {
                int lo, hi;
                need(S.dep[d], S.g.M, mt, lo, hi);
                if (hi >= lo) cnt += hi - lo + 1;
            }
            if (cnt > 64) return AEW_E_UNSUP;
        }
    }
    for (int s = 0; s < n; ++s) {
        // (every producer that SOME stage reads publishes, also through a pruned dependency: the implying stage waits on
        // ITS producers only, so a stage read solely through pruned dependencies must still be complete - it is, by the
        // chain of waits - and it needs no counter.  Stages nobody waits for skip the drain and the atomic.)
        for (int i8 = out[s].first_block >> 3; i8 < (out[s].first_block + out[s].n_blocks) >> 3; ++i8) block_stage[i8] = (uint16_t)s;
    }
    *n_blocks_out = blocks;
    *n_counters_out = counters;
    *set_out = has_dfg || store_win ? 1 : 0;
    return 0;
}

// The same under a caller's tuning record (the record the launches will run under: it decides which stages take the one-window
// body and which launches count as small) instead of the process-wide one.
extern "C" int aew_nt_chain_build_tuned(const aew_gemm_nt_t* descs, int n, aew_nt_stage_t* out, uint16_t* block_stage,
                                        int cap_blocks, int* n_blocks_out, int* n_counters_out, int* set_out, int force,
                                        const aew_tuning_t* tuning) {
    const aew_tuning_t* prev = t_tune;
    if (tuning) t_tune = tuning;
    const int rc = aew_nt_chain_build(descs, n, out, block_stage, cap_blocks, n_blocks_out, n_counters_out, set_out, force);
    t_tune = prev;
    return rc;
}

static int launch_nt_chain(const aew_nt_chain_t& c, hipStream_t st) {
    if (!c.stages || !c.block_stage || !c.counters || c.n_stages < 1 || c.n_blocks < 8 || (c.n_blocks & 7) || c.n_counters < 1 ||
        (c.set != 0 && c.set != 1))
        return AEW_E_ARG;
    // counters + [n_counters] timeout flag, [+1] tiles that waited, [+2] longest wait in polls: zeroed by a kernel of this
    // library in front of every launch (the caller's buffer holds n_counters + 8 words, rounded up to 16 bytes)
    if (!(c.flags & 2)) {                                  // (flags & 2: the caller's plan clears them - one op for all its chains)
        const aew_zero_t z = {c.counters, (int64_t)(((size_t)c.n_counters + 8 + 3) / 4 * 16)};
        const int zr = launch_zero(z, st);
        if (zr) return zr;
    }
    const int spin = c.spin_max > 0 ? c.spin_max : (1 << 18);
    if (c.set == 0)
        hipLaunchKernelGGL((k_nt_chain<0>), dim3(c.n_blocks), dim3(CHAIN_THREADS), CHAIN_LDS_BYTES, st, c.stages, c.block_stage,
                           c.counters, c.n_counters, spin, c.flags, c.sticky);
    else
        hipLaunchKernelGGL((k_nt_chain<1>), dim3(c.n_blocks), dim3(CHAIN_THREADS), CHAIN_LDS_BYTES, st, c.stages, c.block_stage,
                           c.counters, c.n_counters, spin, c.flags, c.sticky);
    return (int)hipGetLastError();
}
