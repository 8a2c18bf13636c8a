// aewavenet.hip — single translation unit of libaewavenet_hip.so: kernels + the C ABI
// declared in include/aewavenet.h.   Build: see __graft_entry__.build()
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC aewavenet.hip -o libaewavenet_hip.so
#include "aew_gemm.hip"
#include "aew_fn.hip"
#include "aew_ops.hip"
#include "aew_sampler.hip"

#include <vector>
#include <cstdlib>

// ---------------------------------------------------------------------------------------------
// timing (HIP events on the plan's stream; used by bench.py for the roofline numbers)
// ---------------------------------------------------------------------------------------------
static int g_timing = 0;
static std::vector<hipEvent_t> g_ev;       // pairs (start, stop)
static std::vector<int32_t> g_ev_tag;
static size_t g_ev_used = 0;

static hipEvent_t ev_get(size_t i) {
    while (g_ev.size() <= i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        g_ev.push_back(e);
    }
    return g_ev[i];
}

static int dispatch(const aew_op_t& op, hipStream_t st) {
    switch (op.kind) {
        case AEW_OP_GEMM_NT: return launch_gemm_nt(op.u.nt, st);
        case AEW_OP_GEMM_TN: return launch_gemm_tn(op.u.tn, st);
        case AEW_OP_COPY_TABLE: return launch_copy(op.u.copy, st);
        case AEW_OP_VQ_NEAREST: return launch_vq_nearest(op.u.vqn, st);
        case AEW_OP_VQ_STATS: return launch_vq_stats(op.u.vqs, st);
        case AEW_OP_VQ_EMA: return launch_vq_ema(op.u.vqe, st);
        case AEW_OP_VQ_BWD: return launch_vq_bwd(op.u.vqb, st);
        case AEW_OP_LC_GATHER: return launch_lc_gather(op.u.lcg, st);
        case AEW_OP_LC_SCATTER: return launch_lc_scatter(op.u.lcs, st);
        case AEW_OP_SPK_BIAS: return launch_spk_bias(op.u.spk, st);
        case AEW_OP_SPK_BWD: return launch_spk_bwd(op.u.spkb, st);
        case AEW_OP_BASE_GATHER: return launch_base_gather(op.u.base, st);
        case AEW_OP_SOFTMAX_NLL: return launch_softmax(op.u.sm, st);
        case AEW_OP_COLSUM: return launch_colsum(op.u.cs, st);
        case AEW_OP_REDUCE: return launch_reduce(op.u.red, st);
        case AEW_OP_ADAM: return launch_adam(op.u.adam, st);
        case AEW_OP_ZERO: return launch_zero(op.u.zero, st);
        case AEW_OP_VAE: return launch_vae(op.u.vae, st);
        case AEW_OP_AE_NORM: return launch_ae_norm(op.u.aen, st);
        case AEW_OP_JITTER: return launch_jitter(op.u.jit, st);
        case AEW_OP_VQ_DIAG: return launch_vq_diag(op.u.diag, st);
        case AEW_OP_MFCC: return launch_mfcc(op.u.mfcc, st);
        case AEW_OP_MOMENTS: return launch_moments(op.u.mom, st);
        case AEW_OP_GEMM_TN_GROUP: return launch_gemm_tn_group(op.u.tng, st);
        case AEW_OP_NT_CHAIN: return 0;      // chaining off / timing mode: the stage ops that follow run one by one (run_ops)
        default: return AEW_E_UNSUP;
    }
}

extern "C" int aew_abi_version(void) { return AEW_ABI_VERSION; }

extern "C" int aew_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(aew_op_t);
        case 1: return (int)sizeof(aew_gemm_nt_t);
        case 2: return (int)sizeof(aew_gemm_tn_t);
        case 3: return (int)sizeof(aew_seg_t);
        case 4: return (int)sizeof(aew_view_t);
        case 5: return (int)sizeof(aew_copy_rec_t);
        case 6: return (int)sizeof(aew_actor_t);
        case 7: return (int)sizeof(aew_sampler_t);
        case 8: return (int)sizeof(aew_tuning_t);
        case 9: return (int)sizeof(aew_nt_stage_t);
        case 10: return (int)sizeof(aew_nt_chain_t);
        default: return -1;
    }
}

// ---------------------------------------------------------------------------------------------
// plan execution.  Lane-1 ops go to a private side stream; fork/join edges are HIP events, which
// inside a stream capture become graph dependencies (the side stream joins the capture by
// waiting on an event recorded in the capturing stream).
// ---------------------------------------------------------------------------------------------
// Default OFF (round 3): with the weight gradients in one launch after the chain there is little left to overlap, and in a
// captured graph every fork / join edge is a cross-branch dependency the replay pays for - measured on one box, one
// process: graph + lanes 6.96 ms/step, graph serial 6.81, eager serial 6.80, eager + lanes 6.77 (profiles/r03_notes.md).
static std::vector<hipEvent_t> g_lane_ev;
static size_t g_lane_ev_next = 0;

extern "C" int aew_set_tn_cursor(int epoch, int slack) {
    if (epoch > 0 && (epoch < 3 || epoch > 64 || slack < 1 || slack > 8)) return AEW_E_ARG;
    g_tune.tn_cursor_epoch = epoch < 0 ? -1 : epoch;
    g_tune.tn_cursor_slack = epoch > 0 ? slack : 0;
    return 0;
}
extern "C" int aew_set_lanes(int on) { g_tune.lanes = on < 0 ? 0 : (on > 2 ? 2 : on); return 0; }   // 2: lanes 4, 5 only (independent chains)

static hipEvent_t lane_event() {
    const size_t POOL = 64;
    if (g_lane_ev.size() < POOL) {
        hipEvent_t e;
        // device-scope release: an event recorded for a wait on another stream of the SAME device must not flush to
        // system scope (AEW_LANE_EVENT_FLAGS: A/B aid, profiles/r04_notes.md §16)
        static const char* fl = getenv("AEW_LANE_EVENT_FLAGS");
        const unsigned flags = fl ? (unsigned)strtoul(fl, nullptr, 0) : (hipEventDisableTiming | hipEventReleaseToDevice);
        if (hipEventCreateWithFlags(&e, flags) != hipSuccess) return nullptr;
        g_lane_ev.push_back(e);
        return e;
    }
    return g_lane_ev[g_lane_ev_next++ % POOL];
}

static int edge(hipStream_t from, hipStream_t to) {          // `to` continues after `from`'s work so far
    hipEvent_t e = lane_event();
    if (!e) return (int)hipErrorOutOfMemory;
    hipError_t rc = hipEventRecord(e, from);
    if (rc != hipSuccess) return (int)rc;
    return (int)hipStreamWaitEvent(to, e, 0);
}

#define AEW_MAX_SIDE 5
static hipStream_t g_side[AEW_MAX_SIDE + 1] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // [1..AEW_MAX_SIDE]

static int run_ops(const aew_op_t* ops, int n, hipStream_t st, int* fail_index, int timing) {   // timing: g_timing
    const bool lanes = AEW_T().lanes && !timing;
    bool main_ahead[AEW_MAX_SIDE + 1];   // main has work side stream k has not been ordered after
    bool open[AEW_MAX_SIDE + 1];         // side stream k has work that main (or a joining side op) has not waited for
    for (int k = 0; k <= AEW_MAX_SIDE; ++k) { main_ahead[k] = true; open[k] = false; }
    int rc = 0;
    for (int i = 0; i < n && rc == 0; ++i) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        hipStream_t target = st;
        const int lane = ops[i].lane;
        if (lane < 0 || lane > AEW_MAX_SIDE) { rc = AEW_E_ARG; if (fail_index) *fail_index = i; break; }
        if (lanes && lane >= (AEW_T().lanes == 2 ? 4 : 1)) {
            if (!g_side[lane]) {
                hipError_t e = hipStreamCreateWithFlags(&g_side[lane], hipStreamNonBlocking);
                if (e != hipSuccess) { rc = (int)e; break; }
            }
            if (main_ahead[lane]) { rc = edge(st, g_side[lane]); main_ahead[lane] = false; }
            if (ops[i].join)                                  // side join: after the other side lanes too
                for (int k = 1; k <= AEW_MAX_SIDE && rc == 0; ++k)
                    if (k != lane && open[k]) rc = edge(g_side[k], g_side[lane]);
            target = g_side[lane];
            open[lane] = true;
        } else {
            if (ops[i].join >= 10 && ops[i].join <= 10 + AEW_MAX_SIDE) {      // one side lane only
                const int k = ops[i].join - 10;
                if (k >= 1 && open[k]) { rc = edge(g_side[k], st); open[k] = false; }
            } else if (ops[i].join)
                for (int k = 1; k <= AEW_MAX_SIDE && rc == 0; ++k)
                    if (open[k]) { rc = edge(g_side[k], st); open[k] = false; }
            for (int k = 1; k <= AEW_MAX_SIDE; ++k) main_ahead[k] = true;
        }
        if (rc != 0) { if (fail_index) *fail_index = i; break; }
        if (timing) {
            e0 = ev_get(2 * g_ev_used);
            e1 = ev_get(2 * g_ev_used + 1);
            if (!e0 || !e1) return (int)hipErrorOutOfMemory;
            (void)hipEventRecord(e0, st);
        }
        int skip = 0;
        if (ops[i].kind == AEW_OP_NT_CHAIN) {
            // the chain's stage ops follow it in the plan: launched as ONE kernel here (and skipped), or - chaining off,
            // per-op timing - left to run one by one
            const aew_nt_chain_t& c = ops[i].u.chain;
            if (c.n_ops < 1 || i + c.n_ops > n - 1) { rc = AEW_E_ARG; if (fail_index) *fail_index = i; break; }
            for (int k = 1; k <= c.n_ops && rc == 0; ++k)
                if (ops[i + k].kind != AEW_OP_GEMM_NT || ops[i + k].lane != lane || (k > 1 && ops[i + k].join)) rc = AEW_E_ARG;
            if (rc != 0) { if (fail_index) *fail_index = i; break; }
            // (a record whose one-window limit differs from the one the stage table was built under would make the chain run
            // other kernels - another summation order - than the stage ops: those then run one by one)
            if (timing != 1 && AEW_T().nt_chain && c.stages && c.built_window == AEW_T().nt_window) {   // (timing 2: the chain is timed as ONE op)
                rc = ensure_big_lds();
                if (rc == 0) rc = launch_nt_chain(c, target);
                skip = c.n_ops;
            }
        } else
            rc = dispatch(ops[i], target);
        if (timing) {
            (void)hipEventRecord(e1, st);
            if (g_ev_tag.size() <= g_ev_used) g_ev_tag.resize(g_ev_used + 1);
            g_ev_tag[g_ev_used] = ops[i].tag;
            ++g_ev_used;
        }
        if (rc != 0 && fail_index) *fail_index = i;
        for (int k = 0; k < skip && timing; ++k) {            // the chain's stage ops: empty intervals, so that event index = op index
            hipEvent_t a = ev_get(2 * g_ev_used), b2 = ev_get(2 * g_ev_used + 1);
            if (!a || !b2) return (int)hipErrorOutOfMemory;
            (void)hipEventRecord(a, st);
            (void)hipEventRecord(b2, st);
            if (g_ev_tag.size() <= g_ev_used) g_ev_tag.resize(g_ev_used + 1);
            g_ev_tag[g_ev_used] = ops[i + 1 + k].tag;
            ++g_ev_used;
        }
        i += skip;
    }
    for (int k = 1; k <= AEW_MAX_SIDE; ++k)                  // implicit join (also on the error path,
        if (open[k]) {                                       // so a capture is never left forked)
            const int jr = edge(g_side[k], st);
            if (rc == 0) rc = jr;
        }
    return rc;
}

extern "C" int aew_run_plan(const aew_op_t* ops, int n, void* stream, int* fail_index) {
    if (!ops || n < 0) return AEW_E_ARG;
    return run_ops(ops, n, (hipStream_t)stream, fail_index, g_timing);
}

// ---- tuning context: a caller's own record for the duration of one call (thread-local), see aewavenet.h
struct TuneScope {
    const aew_tuning_t* prev;
    explicit TuneScope(const aew_tuning_t* t) : prev(t_tune) { if (t) t_tune = t; }
    ~TuneScope() { t_tune = prev; }
};
static void tune_clamp(aew_tuning_t& t) {
    auto cl = [](int32_t& v, int lo, int hi) { v = v < lo ? lo : (v > hi ? hi : v); };
    if (t.nt_wave_rows != 0 && t.nt_wave_rows != 1 && t.nt_wave_rows != 64 && t.nt_wave_rows != 128 && t.nt_wave_rows != 256 &&
        t.nt_wave_rows != 512)
        t.nt_wave_rows = 64;
    cl(t.nt_pipe, 0, 2); cl(t.nt_rows192, 0, 2); cl(t.nt_window, 0, 64); cl(t.nt_mem128, 0, 2); cl(t.nt_deep, 0, 3);
    cl(t.lanes, 0, 2); if (t.tn_cursor_epoch > 0) { cl(t.tn_cursor_epoch, 3, 64); cl(t.tn_cursor_slack, 1, 8); } else if (t.tn_cursor_epoch < 0) t.tn_cursor_epoch = -1; cl(t.nt_small_w8, 0, 1); cl(t.nt_chain, 0, 1); cl(t.deterministic, 0, 1); cl(t.tn_mfma32, 0, 1); cl(t.nf_loaders, 0, 1); cl(t.fn_enable, 0, 1); cl(t.tn_safe, 0, 1); cl(t.tn_big, 0, 1);
    cl(t.nt_small_tiles, 0, 1 << 30); cl(t.nt_small_n64, 0, 1 << 30); cl(t.nt_small_deep, 0, 1 << 30); cl(t.nf_deep, 0, 1 << 30);
    cl(t.fn_ring3, 0, 1 << 30); cl(t.tn_big_target, 1, 1 << 30); cl(t.tn_fold_rows, 0, 1 << 30); cl(t.tn_target_blocks, 1, 1 << 30);
    cl(t.tn_small_tiles, 0, 1 << 30); cl(t.tn_small_target, 1, 1 << 30);
}
extern "C" int aew_tuning_default(aew_tuning_t* out) {
    if (!out) return AEW_E_ARG;
    static const aew_tuning_t d = AEW_TUNING_DEFAULTS;
    *out = d;
    return 0;
}
extern "C" int aew_tuning_get(aew_tuning_t* out) {
    if (!out) return AEW_E_ARG;
    *out = g_tune;
    return 0;
}
extern "C" int aew_tuning_set(const aew_tuning_t* in) {
    if (!in) return AEW_E_ARG;
    aew_tuning_t t = *in;
    tune_clamp(t);
    g_tune = t;
    return 0;
}
// The split-K plan of a TN op (slabs, rows per split, batch fold) is fixed when a PLAN IS BUILT: aew_tn_slabs() sizes the slab
// buffers and the unpack tables under the process-wide record.  A caller's per-call record must therefore not change it at
// launch time - more slabs than allocated would be written out of bounds, fewer would leave stale ones in the sum - so the
// fields that feed tn_plan() always come from the process-wide record.
static void tune_pin_tn_plan(aew_tuning_t& t) {
    t.tn_fold_rows = g_tune.tn_fold_rows; t.tn_target_blocks = g_tune.tn_target_blocks;
    t.tn_small_tiles = g_tune.tn_small_tiles; t.tn_small_target = g_tune.tn_small_target;
    t.tn_big = g_tune.tn_big; t.tn_big_target = g_tune.tn_big_target;
}
extern "C" int aew_run_plan_tuned(const aew_op_t* ops, int n, void* stream, int* fail_index, const aew_tuning_t* tuning) {
    if (!ops || n < 0) return AEW_E_ARG;
    aew_tuning_t t;
    if (tuning) { t = *tuning; tune_clamp(t); tune_pin_tn_plan(t); }
    TuneScope scope(tuning ? &t : nullptr);
    return run_ops(ops, n, (hipStream_t)stream, fail_index, g_timing);
}
extern "C" int aew_graph_capture(const aew_op_t* ops, int n, void** exec_out, int* fail_index);
extern "C" int aew_graph_capture_tuned(const aew_op_t* ops, int n, void** exec_out, int* fail_index, const aew_tuning_t* tuning) {
    aew_tuning_t t;
    if (tuning) { t = *tuning; tune_clamp(t); tune_pin_tn_plan(t); }
    TuneScope scope(tuning ? &t : nullptr);
    return aew_graph_capture(ops, n, exec_out, fail_index);
}

// ---------------------------------------------------------------------------------------------
// hipGraph capture of a plan: the ops are recorded once on a private capture stream and
// replayed with one hipGraphLaunch per step (removes the per-kernel host launch gaps).
// ---------------------------------------------------------------------------------------------
static hipStream_t g_cap_stream = nullptr;

extern "C" int aew_graph_capture(const aew_op_t* ops, int n, void** exec_out, int* fail_index) {
    if (!ops || n <= 0 || !exec_out) return AEW_E_ARG;
    int rc = ensure_big_lds();                       // attribute calls are not capturable
    if (rc) return rc;
    hipError_t e;
    if (!g_cap_stream) {
        e = hipStreamCreateWithFlags(&g_cap_stream, hipStreamNonBlocking);
        if (e != hipSuccess) return (int)e;
    }
    e = hipStreamBeginCapture(g_cap_stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) return (int)e;
    rc = run_ops(ops, n, g_cap_stream, fail_index, 0);
    hipGraph_t graph = nullptr;
    e = hipStreamEndCapture(g_cap_stream, &graph);
    if (rc != 0) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return (int)e;
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return (int)e;
    *exec_out = (void*)exec;
    return 0;
}

extern "C" int aew_graph_launch(void* exec, void* stream) {
    if (!exec) return AEW_E_ARG;
    return (int)hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream);
}

extern "C" int aew_graph_destroy(void* exec) {
    if (!exec) return 0;
    return (int)hipGraphExecDestroy((hipGraphExec_t)exec);
}

extern "C" int aew_timing_enable(int on) {
    g_timing = on == 2 ? 2 : (on ? 1 : 0);       // 2: chained launches (AEW_OP_NT_CHAIN) are timed as ONE op instead of stage by stage
    g_ev_used = 0;
    return 0;
}

extern "C" int aew_timing_read(float* ms, int32_t* tags, int capacity, int* count) {
    if (!count) return AEW_E_ARG;
    *count = (int)g_ev_used;
    if (g_ev_used == 0) return 0;
    hipError_t e = hipEventSynchronize(g_ev[2 * g_ev_used - 1]);
    if (e != hipSuccess) return (int)e;
    for (size_t i = 0; i < g_ev_used && (int)i < capacity; ++i) {
        float t = 0.f;
        e = hipEventElapsedTime(&t, g_ev[2 * i], g_ev[2 * i + 1]);
        if (e != hipSuccess) return (int)e;
        if (ms) ms[i] = t;
        if (tags) tags[i] = g_ev_tag[i];
    }
    g_ev_used = 0;
    return 0;
}

extern "C" int aew_set_tn_safe(int on) { g_tune.tn_safe = on ? 1 : 0; return 0; }
extern "C" int aew_set_tn_big(int on, int target_blocks) {
    g_tune.tn_big = on ? 1 : 0;
    if (target_blocks > 0) g_tune.tn_big_target = target_blocks;
    return 0;
}
extern "C" int aew_set_tn_small(int max_tiles, int target_blocks) {
    if (max_tiles < 0 || target_blocks < 1) return AEW_E_ARG;
    g_tune.tn_small_tiles = max_tiles;
    g_tune.tn_small_target = target_blocks;
    return 0;
}
extern "C" int aew_set_tn_target_blocks(int n) {
    if (n < 1) return AEW_E_ARG;
    g_tune.tn_target_blocks = n;
    return 0;
}
extern "C" int aew_sampler_run(const aew_sampler_t* s, void* stream) {
    if (!s) return AEW_E_ARG;
    return launch_sampler(*s, reinterpret_cast<hipStream_t>(stream));
}
extern "C" int aew_set_nt_rows192(int mode) { g_tune.nt_rows192 = mode < 0 ? 0 : (mode > 2 ? 2 : mode); return 0; }
extern "C" int aew_set_nt_mem128(int mode) { g_tune.nt_mem128 = mode < 0 ? 0 : (mode > 2 ? 2 : mode); return 0; }
extern "C" int aew_set_nt_deep(int mode) { g_tune.nt_deep = mode < 0 ? 0 : (mode > 3 ? 3 : mode); return 0; }
extern "C" int aew_set_nt_small_tiles(int n) { g_tune.nt_small_tiles = n < 0 ? 0 : n; return 0; }
extern "C" int aew_set_nf_loaders(int on) { g_tune.nf_loaders = on ? 1 : 0; return 0; }
extern "C" int aew_set_nf_deep(int max_blocks) { g_tune.nf_deep = max_blocks < 0 ? 0 : max_blocks; return 0; }
extern "C" int aew_set_nt_small_waves(int waves) { g_tune.nt_small_w8 = waves >= 8 ? 1 : 0; return 0; }
extern "C" int aew_set_nt_small_n64(int max_blocks) { g_tune.nt_small_n64 = max_blocks < 0 ? 0 : max_blocks; return 0; }
extern "C" int aew_set_nt_small_deep(int max_blocks) { g_tune.nt_small_deep = max_blocks < 0 ? 0 : max_blocks; return 0; }
extern "C" int aew_set_nt_pipe(int mode) { g_tune.nt_pipe = mode < 0 ? 0 : (mode > 2 ? 2 : mode); return 0; }
extern "C" int aew_set_nt_wave_rows(int rows) {
    if (rows != 0 && rows != 1 && rows != 64 && rows != 128 && rows != 256 && rows != 512) return AEW_E_ARG;
    g_tune.nt_wave_rows = rows;
    return 0;
}

extern "C" const char* aew_strerror(int code) {
    if (code == 0) return "ok";
    if (code == AEW_E_ARG) return "aewavenet: malformed descriptor";
    if (code == AEW_E_UNSUP) return "aewavenet: unsupported op/dtype/flag combination";
    if (code == AEW_E_ALIGN) return "aewavenet: pointer/pitch alignment violated";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "aewavenet: unknown error";
}

// ---------------------------------------------------------------------------------------------
// device self-test of the lane mappings the kernels rely on
//   detail[0] bf16 MFMA 16x16x32 C/D map + operand map        (0 ok)
//   detail[1] f32  MFMA 16x16x4  C/D map                       (0 ok)
//   detail[2] f32  MFMA is a k-ascending fmaf chain (bitwise)  (0 ok)
//   detail[3] ds_read_b64_tr_b16 lane semantics                (0 ok)
//   detail[4] global_load_lds_dwordx4 lands lane-linear        (0 ok)
// ---------------------------------------------------------------------------------------------
__global__ void k_selftest(int32_t* detail, float* scratch) {
    __shared__ __attribute__((aligned(16))) char lds[8192];
    const int lane = threadIdx.x;
    const int fi = lane & 15, fg = lane >> 4;
    int bad;
    // ---- [0] bf16 MFMA: A[i][k] = small ints asymmetric, B[k][j]
    {
        bf16x8_t a, bq;
        for (int e = 0; e < 8; ++e) {
            const int k = 8 * fg + e;
            a[e] = (__bf16)(float)((fi * 3 + k) % 7 - 3);            // A[fi][k]
            bq[e] = (__bf16)(float)((fi * 5 + 2 * k) % 5 - 2);       // B[k][fi]
        }
        f32x4_t c = {0, 0, 0, 0};
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq, c, 0, 0, 0);
        bad = 0;
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * fg + r, col = fi;
            float ref = 0.f;
            for (int k = 0; k < 32; ++k)
                ref += (float)((row * 3 + k) % 7 - 3) * (float)((col * 5 + 2 * k) % 5 - 2);
            if (c[r] != ref) bad = 1;
        }
        if (__any(bad) && lane == 0) detail[0] = 1;
    }
    // ---- [1]/[2] f32 MFMA
    {
        f32x4_t c = {0, 0, 0, 0};
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 4 * ks + fg;
            const float a = (float)((fi * 3 + k) % 7 - 3), bq = (float)((fi * 5 + 2 * k) % 5 - 2);
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bq, c, 0, 0, 0);
        }
        bad = 0;
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * fg + r, col = fi;
            float ref = 0.f;
            for (int k = 0; k < 16; ++k)
                ref += (float)((row * 3 + k) % 7 - 3) * (float)((col * 5 + 2 * k) % 5 - 2);
            if (c[r] != ref) bad = 1;
        }
        if (__any(bad) && lane == 0) detail[1] = 1;
        // order sensitivity: values with wide dynamic range, compare bitwise with fmaf chain
        f32x4_t d = {0.25f, 0.25f, 0.25f, 0.25f};
        // exactly representable operands built without floating-point arithmetic: odd integers
        // scaled by powers of two spread over 40 binades, so partial sums round differently
        // under any other summation order
        auto av = [](int i, int k) { return ldexpf((float)(((i * 37 + k * 11) % 97) * 2 + 1) * ((k & 1) ? -1.f : 1.f), ((k * 7) % 5) * 5 - 10); };
        auto bv = [](int k, int j) { return ldexpf((float)(((j * 29 + k * 13) % 89) * 2 + 1), ((k * 3) % 4) * 5 - 10); };
        for (int ks = 0; ks < 8; ++ks) {
            const int k = 4 * ks + fg;
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(av(fi, k), bv(k, fi), d, 0, 0, 0);
        }
        bad = 0;
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * fg + r, col = fi;
            float ref = 0.25f;
            for (int k = 0; k < 32; ++k) ref = __fmaf_rn(av(row, k), bv(k, col), ref);
            if (__float_as_uint(d[r]) != __float_as_uint(ref)) bad = 1;
        }
        if (__any(bad) && lane == 0) detail[2] = 1;
    }
    // ---- [3] ds_read_b64_tr_b16: fill 4 rows x 64 cols of shorts with value row*64+col
    {
        short* t = reinterpret_cast<short*>(lds);
        for (int e = lane; e < 4 * 64 * 4; e += 64) t[e] = (short)e;      // 16 rows x 64 cols
        __syncthreads();
        // lane (q, g): read rows 4g..4g+3, column group: address row (4g + (q>>2)), cols 4*(q&3)
        const int q = lane & 15;
        const char* p = lds + ((4 * fg + (q >> 2)) * 64 + 4 * (q & 3)) * 2;
        const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4_t __attribute__((address_space(3)))*)AEW_LDS_PTR(p));
        bad = 0;
        for (int e = 0; e < 4; ++e)
            if (v[e] != (short)((4 * fg + e) * 64 + q)) bad = 1;          // rows 4g+e, column q
        if (__any(bad) && lane == 0) detail[3] = 1;
        __syncthreads();
    }
    // ---- [4] LDS-DMA placement
    {
        float* src = scratch;                                           // 256 floats prepared by host memset
        for (int e = lane; e < 256; e += 64) src[e] = (float)e;
        __threadfence();
        __syncthreads();
        // lane l fetches 16 B at element 4*(63-l)  -> LDS slot l
        glds16(src + 4 * (63 - lane), lds);
        wait_vm0();
        __syncthreads();
        const float* t = reinterpret_cast<const float*>(lds);
        bad = 0;
        for (int e = 0; e < 4; ++e)
            if (t[4 * lane + e] != (float)(4 * (63 - lane) + e)) bad = 1;
        if (__any(bad) && lane == 0) detail[4] = 1;
    }
}

// ---------------------------------------------------------------------------------------------
// box fingerprint (bench.py's `box` record): what THIS device sustains on two primitives, measured with HIP events on the
// caller's stream right before a benchmark, so that two bench lines from two boxes of the pool can be told apart from
// two builds (the pool's boxes differ by +-0.15 ms per step on one binary; profiles/r02_notes.md).
//   out[0] TFLOP/s of a pure v_mfma_f32_16x16x32_bf16 loop (512 blocks x 512 threads, 16 accumulators: the NT kernels' wave shape)
//   out[1] TB/s (read + written) of a device-to-device copy of `copy_bytes` with 16-byte accesses
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_probe_mfma(float* out, int iters) {
    f32x4_t acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_probe_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

extern "C" int aew_probe_box(void* scratch, int64_t scratch_bytes, int64_t copy_bytes, void* stream, float* out) {
    if (!scratch || !out || copy_bytes < (1 << 20) || (copy_bytes & 15) || scratch_bytes < 2 * copy_bytes ||
        scratch_bytes < 512 * 512 * 4 || ((uintptr_t)scratch & 15))
        return AEW_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return (int)hipErrorOutOfMemory;
    float ms = 0.f;
    int rc = 0;
    const int iters = 2000, reps = 5;
    hipLaunchKernelGGL(k_probe_mfma, dim3(512), dim3(512), 0, st, reinterpret_cast<float*>(scratch), iters);       // warm-up
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_probe_mfma, dim3(512), dim3(512), 0, st, reinterpret_cast<float*>(scratch), iters);
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = (int)hipGetLastError();
    out[0] = ms > 0.f ? (float)((double)512 * 8 * iters * 16 * reps * (16.0 * 16 * 32 * 2) / (ms * 1e-3) / 1e12) : 0.f;
    const uint4* src = reinterpret_cast<const uint4*>(scratch);
    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(scratch) + copy_bytes);
    const int64_t n16 = copy_bytes / 16;
    hipLaunchKernelGGL(k_probe_copy, dim3(4096), dim3(256), 0, st, src, dst, n16);
    (void)hipEventRecord(e0, st);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k_probe_copy, dim3(4096), dim3(256), 0, st, src, dst, n16);
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) rc = (int)hipGetLastError();
    out[1] = ms > 0.f ? (float)(2.0 * (double)copy_bytes * reps / (ms * 1e-3) / 1e12) : 0.f;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc == 0) rc = (int)hipGetLastError();
    return rc;
}

extern "C" int aew_selftest(void* scratch, int64_t scratch_bytes, void* stream, int32_t* detail) {
    if (!scratch || scratch_bytes < (1 << 16) || !detail) return AEW_E_ARG;
    hipStream_t st = (hipStream_t)stream;
    int32_t* d_detail = reinterpret_cast<int32_t*>(scratch);
    float* d_f = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + 4096);
    hipError_t e = hipMemsetAsync(d_detail, 0, 8 * sizeof(int32_t), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(k_selftest, dim3(1), dim3(64), 0, st, d_detail, d_f);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    e = hipMemcpyAsync(detail, d_detail, 8 * sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return (int)e;
    e = hipStreamSynchronize(st);
    if (e != hipSuccess) return (int)e;
    for (int i = 0; i < 5; ++i)
        if (detail[i]) return AEW_E_UNSUP;
    return 0;
}
