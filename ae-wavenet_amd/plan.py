"""Launch-plan infrastructure: workspace buffers, ctypes op emission, plan execution.

A *plan* is a ctypes array of `aew_op_t` (include/aewavenet.h) that `aew_run_plan` executes
in order on one HIP stream.  Plans are built once per (model, batch, window) against a
persistent :class:`Workspace`, so a training step is a handful of C calls with no
per-step Python work proportional to the number of kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L

TORCH_DT = {L.BF16: torch.bfloat16, L.F32: torch.float32}
ESIZE = {L.BF16: 2, L.F32: 4}


def ru(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class Workspace:
    """Named device buffers.  Every buffer is a flat torch tensor; plans hold raw pointers
    into them, so buffers are never reallocated once a plan references them."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.bufs: Dict[str, torch.Tensor] = {}

    def alloc(self, name: str, numel: int, dtype, zero: bool = True) -> torch.Tensor:
        if name in self.bufs:
            raise KeyError(f"buffer {name} already allocated")
        numel = max(int(numel), 8)
        # +64 elements of slack so 16-byte vector accesses at the tail stay in bounds
        t = (torch.zeros if zero else torch.empty)(numel + 64, dtype=dtype, device=self.device)
        self.bufs[name] = t
        return t

    def get(self, name: str) -> torch.Tensor:
        return self.bufs[name]

    def ptr(self, name: str, off: int = 0) -> int:
        t = self.bufs[name]
        return t.data_ptr() + off * t.element_size()

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.bufs.values())

    def resolve(self, addr: int) -> Tuple[str, int]:
        """(buffer name, element offset) containing device address `addr` (used by the CPU
        plan interpreter in tests/)."""
        for n, t in self.bufs.items():
            p = t.data_ptr()
            if p <= addr < p + t.numel() * t.element_size():
                return n, (addr - p) // t.element_size()
        raise KeyError(f"address {addr:#x} is not inside any workspace buffer")


class Mat:
    """A channels-last matrix [batch][rows][pitch] inside a workspace buffer."""

    def __init__(self, ws: Workspace, name: str, batch: int, rows: int, pitch: int, dtype: int,
                 cols: Optional[int] = None, base_off: int = 0, bs: Optional[int] = None):
        self.ws, self.name, self.batch, self.rows, self.pitch, self.dtype = ws, name, batch, rows, pitch, dtype
        self.cols = pitch if cols is None else cols
        self.base_off = base_off
        self.bs = rows * pitch if bs is None else bs

    @staticmethod
    def new(ws: Workspace, name: str, batch: int, rows: int, pitch: int, dtype: int,
            cols: Optional[int] = None, guard: int = 0) -> "Mat":
        """guard: that many scratch rows before and after every batch element's rows (a writer whose
        GEMM row covers several matrix rows may spill into them; readers never see them)."""
        ws.alloc(name, batch * (rows + 2 * guard) * pitch, TORCH_DT[dtype])
        return Mat(ws, name, batch, rows, pitch, dtype, cols, base_off=guard * pitch, bs=(rows + 2 * guard) * pitch)

    @property
    def ptr(self) -> int:
        return self.ws.ptr(self.name, self.base_off)

    def tensor(self) -> torch.Tensor:
        t = self.ws.get(self.name)
        return torch.as_strided(t, (self.batch, self.rows, self.pitch), (self.bs, self.pitch, 1), self.base_off)

    def seg(self, k_len: int, row_off: int = 0, row_step: int = 1, lo: int = 0,
            hi: Optional[int] = None, col_off: int = 0, b0: int = 0) -> L.Seg:
        """b0: first batch element (ops that cover a sub-range of the batch)."""
        s = L.Seg()
        s.ptr = self.ptr + (col_off + b0 * self.bs) * ESIZE[self.dtype]
        s.batch_stride, s.row_pitch = self.bs, self.pitch
        s.row_step, s.row_off = row_step, row_off
        s.row_lo, s.row_hi = lo, self.rows if hi is None else hi
        s.k_len = k_len
        return s

    def view(self, row_off: int = 0, row_step: int = 1, lo: int = 0, hi: Optional[int] = None,
             col_off: int = 0, b0: int = 0) -> L.View:
        v = L.View()
        v.ptr = self.ptr + (col_off + b0 * self.bs) * ESIZE[self.dtype]
        v.batch_stride, v.row_pitch = self.bs, self.pitch
        v.row_step, v.row_off = row_step, row_off
        v.row_lo, v.row_hi = lo, self.rows if hi is None else hi
        v.dtype = self.dtype
        return v


def null_view() -> L.View:
    return L.View()


class Plan:
    def __init__(self, name: str):
        self.name = name
        self.ops: List[L.Op] = []
        self.labels: List[str] = []
        self._arr = None
        self._graph = None            # hipGraphExec handle (lazy)
        self.keep: list = []          # device tables etc. that must outlive the plan
        self.lane = 0                 # lane given to ops added next (see aew_op_t in aewavenet.h)

    N_SIDE = 5

    def side(self, lane: int = 1):
        """`with plan.side(k):` — ops added inside are off the critical chain, on side lane k (1..4)."""
        plan = self
        assert 1 <= lane <= Plan.N_SIDE

        class _Side:
            def __enter__(self):
                self.prev, plan.lane = plan.lane, lane

            def __exit__(self, *a):
                plan.lane = self.prev
        return _Side()

    def add(self, kind: int, payload, label: str, tag: int = 0, join=False) -> L.Op:
        """join: False / True (all side lanes) / ("lane", k) = side lane k only (main-lane ops)."""
        op = L.Op()
        # tag = semantic tag + 100 * kernel class (1 NT bf16, 2 TN bf16, 3 NT f32, 4 TN f32)
        cls = 0
        if kind == L.OP_GEMM_NT:
            cls = 1 if payload.dtype == L.BF16 else 3
        elif kind == L.OP_GEMM_TN:
            cls = 2 if payload.dtype == L.BF16 else 4
        elif kind == L.OP_GEMM_TN_GROUP:
            cls = 2
        op.kind, op.tag = kind, tag + 100 * cls
        op.lane, op.join = self.lane, (10 + join[1] if isinstance(join, tuple) else int(join))
        setattr(op.u, L.OP_FIELD[kind], payload)
        self.ops.append(op)
        self.labels.append(label)
        self._arr = None
        return op

    def zero(self, ws: Workspace, name: str, label: Optional[str] = None):
        z = L.Zero()
        t = ws.get(name)
        z.ptr, z.bytes = t.data_ptr(), (t.numel() * t.element_size()) // 16 * 16
        self.add(L.OP_ZERO, z, label or f"zero:{name}")

    def array(self):
        if self._arr is None:
            self._arr = (L.Op * len(self.ops))(*self.ops)
        return self._arr

    def invalidate_graph(self):
        """Call after mutating an op inside array(): the captured graph froze the old values."""
        if self._graph is not None:
            L.load().aew_graph_destroy(self._graph)
            self._graph = None

    def run_graph(self, stream: int = 0, tuning: Optional["L.Tuning"] = None):
        """Replay the plan as a hipGraph (captured on first use; `tuning` is what the capture runs under - a captured
        graph keeps it, invalidate_graph() to change)."""
        if not self.ops:
            return
        lib = L.load()
        if self._graph is None:
            h = C.c_void_p()
            fail = C.c_int(-1)
            if tuning is not None:
                rc = lib.aew_graph_capture_tuned(C.cast(self.array(), C.c_void_p), len(self.ops), C.byref(h), C.byref(fail),
                                                 C.byref(tuning))
            else:
                rc = lib.aew_graph_capture(C.cast(self.array(), C.c_void_p), len(self.ops), C.byref(h), C.byref(fail))
            if rc != 0:
                lab = self.labels[fail.value] if 0 <= fail.value < len(self.labels) else "?"
                L.check(rc, f"graph capture of plan '{self.name}' (op '{lab}')", fail.value)
            self._graph = h
        L.check(lib.aew_graph_launch(self._graph, C.c_void_p(stream)), f"graph launch '{self.name}'")

    def run(self, stream: int = 0, tuning: Optional["L.Tuning"] = None):
        """tuning: an aew_tuning_t for THIS call only (kernel shapes; the process-wide switches are not touched)."""
        if not self.ops:
            return
        fail = C.c_int(-1)
        if tuning is not None:
            rc = L.load().aew_run_plan_tuned(C.cast(self.array(), C.c_void_p), len(self.ops), C.c_void_p(stream),
                                             C.byref(fail), C.byref(tuning))
        else:
            rc = L.load().aew_run_plan(C.cast(self.array(), C.c_void_p), len(self.ops), C.c_void_p(stream),
                                       C.byref(fail))
        if rc != 0:
            lab = self.labels[fail.value] if 0 <= fail.value < len(self.labels) else "?"
            L.check(rc, f"plan '{self.name}' op '{lab}'", fail.value)


# ------------------------------------------------------------------------------------------
# GEMM op construction helpers
# ------------------------------------------------------------------------------------------
def make_nt(dtype: int, M: int, N: int, N_pad: int, batch: int, segs: Sequence[L.Seg], W_ptr: int,
            epi: int = L.EPI_STORE, flags: int = 0, out0: Optional[L.View] = None,
            out1: Optional[L.View] = None, out2: Optional[L.View] = None,
            aux0: Optional[L.View] = None, aux1: Optional[L.View] = None, bias_ptr: int = 0,
            bias_bs: int = 0, n_split: int = 0, counter_ptr: int = 0, impl: int = 0,
            W2_ptr: int = 0, N2: int = 0, N2_pad: int = 0, out3: Optional[L.View] = None,
            k_split: int = 0, ksplit_ws_ptr: int = 0, ksplit_tickets_ptr: int = 0) -> L.GemmNT:
    g = L.GemmNT()
    g.dtype, g.impl, g.M, g.N, g.N_pad, g.batch = dtype, impl, M, N, N_pad, batch
    if not 1 <= len(segs) <= L.MAX_SEGS:
        raise ValueError(f"{len(segs)} segments (max {L.MAX_SEGS})")
    g.n_segs = len(segs)
    for i, s in enumerate(segs):
        g.seg[i] = s
    g.K_total = sum(s.k_len for s in segs)
    g.W = W_ptr
    g.epi, g.flags = epi, flags
    for nm, v in (("out0", out0), ("out1", out1), ("out2", out2), ("aux0", aux0), ("aux1", aux1)):
        if v is not None:
            setattr(g, nm, v)
    g.bias, g.bias_bs, g.n_split = bias_ptr or None, bias_bs, n_split
    g.counter = counter_ptr or None
    if k_split > 1:                  # exact fp32 kernel: S contiguous k-ranges, fixed-order combine (aewavenet.h)
        g.k_split, g.ksplit_ws, g.ksplit_tickets = k_split, ksplit_ws_ptr, ksplit_tickets_ptr
    if W2_ptr:                       # fused gated layer: residual 1x1 over the z tile (aewavenet.h)
        g.W2, g.N2, g.N2_pad = W2_ptr, N2, N2_pad
        g.out3 = out3
    return g


def make_tn(dtype: int, Mc: int, batch: int, N: int, N_pad: int, gseg: L.Seg, segs: Sequence[L.Seg],
            impl: int = 0) -> L.GemmTN:
    t = L.GemmTN()
    t.dtype, t.impl, t.Mc, t.batch, t.N, t.N_pad = dtype, impl, Mc, batch, N, N_pad
    t.g = gseg
    if not 1 <= len(segs) <= L.MAX_SEGS:
        raise ValueError(f"{len(segs)} segments (max {L.MAX_SEGS})")
    t.n_segs = len(segs)
    for i, s in enumerate(segs):
        t.seg[i] = s
    t.K_total = sum(s.k_len for s in segs)
    t.snap_k = -1
    return t


class TnGroupBuilder:
    """Collects bf16 TN descriptors and emits them as ONE grouped launch (AEW_OP_GEMM_TN_GROUP): every output tile
    contracts over the whole time axis and all batch elements in one block, so each descriptor gets one result (no
    split-K slabs).  Descriptors and tile map are uploaded to device memory at emit()."""

    N_XCD = 8
    CURSOR_AUTO_TILES = 1024              # (the 256 CUs hold 768 of the 128-tile blocks at a time)

    def __init__(self, ws: Workspace, name: str, tile: int = 128):
        assert tile in (128, 256, 384)    # 384: the 8-wave 128 x 256 / 256 x 128 tiles (k_gemm_tn_bf16_grp8; aewavenet.h)
        self.ws, self.name, self.tile = ws, name, tile
        self.descs: List[L.GemmTN] = []
        self.labels: List[str] = []
        self.cursor = False               # True: emit() provides the row cursor's progress words (and the op that zeroes them)
                                          # = the launch is paced; None: iff the group has more tiles than CURSOR_AUTO_TILES

    def _grid(self, t: L.GemmTN) -> Tuple[int, int]:
        """(k tiles, n tiles) of a descriptor's output in this builder's tile size."""
        k128, n128 = t.K_total // 128, t.N_pad // 128
        if self.tile == 384:              # orientation per descriptor (tn8_ori in csrc/aew_gemm.hip: the same rule)
            ori = 1 if (t.K_total % 256 == 0 and t.N_pad % 256 != 0) else 0
            return ((k128 + 1) // 2, n128) if ori else (k128, (n128 + 1) // 2)
        return (k128, n128) if self.tile == 128 else ((k128 + 1) // 2, (n128 + 1) // 2)

    def set_split(self, t: L.GemmTN, chunk_rows: int) -> int:
        """Split the descriptor's contraction into chunks of about `chunk_rows` rows per batch element, one block and one
        partial-sum slab each (matrices with few output tiles and many rows; 128-tile groups only).  Returns the number
        of slabs `out` must hold (out_batch_stride apart)."""
        if chunk_rows > 0 and self.tile == 128:
            sp = max(1, -(-t.Mc // chunk_rows))
            rows = ru(-(-t.Mc // sp), 32)
            t.grp_splits, t.grp_rows = -(-t.Mc // rows), rows
        return t.grp_splits * t.batch if t.grp_splits > 0 else 1

    def add(self, t: L.GemmTN, label: str):
        L.check(L.load().aew_tn_group_check(C.byref(t)), f"grouped TN descriptor '{label}'")
        self.descs.append(t)
        self.labels.append(label)

    def tile_map(self) -> List[int]:
        """blockIdx -> desc << 22 | chunk << 12 | tile.  Workgroup p runs on XCD p % 8: the tiles of one descriptor (they share
        the G rows along k tiles and the A rows along n tiles) go to one XCD where there are enough descriptors,
        otherwise a descriptor is cut by n tile; longest-first onto the least loaded XCD."""
        units = []                                           # (tiles, desc, [records])
        for d, t in enumerate(self.descs):
            nkt, nnt = self._grid(t)
            if t.grp_splits > 0:                             # split descriptor: the tiles of one (batch, chunk) together
                for c in range(t.grp_splits * t.batch):
                    units.append((nkt * nnt, d, [(d << 22) | (c << 12) | tl for tl in range(nkt * nnt)]))
            elif len(self.descs) >= self.N_XCD:
                units.append((nkt * nnt, d, [(d << 22) | tl for tl in range(nkt * nnt)]))
            else:
                for nt in range(nnt):
                    units.append((nkt, d, [(d << 22) | (nt * nkt + kt) for kt in range(nkt)]))
        lists = [[] for _ in range(self.N_XCD)]
        for n, d, recs in sorted(units, key=lambda u: (-u[0], u[1], u[2][0])):
            x = min(range(self.N_XCD), key=lambda i: (len(lists[i]), i))
            lists[x].extend(recs)
        depth = max(len(l) for l in lists)
        out = []
        for i in range(depth):
            for x in range(self.N_XCD):
                out.append(lists[x][i] if i < len(lists[x]) else -1)
        return out

    def emit(self, plan: "Plan", label: str, tag: int = 0, join: bool = False):
        if not self.descs:
            return None
        raw = bytes((L.GemmTN * len(self.descs))(*self.descs))
        dt = self.ws.alloc(f"{self.name}.descs", (len(raw) + 7) // 8, torch.int64, zero=True)
        dt[:(len(raw) + 7) // 8].copy_(torch.frombuffer(bytearray(raw + b"\0" * (-len(raw) % 8)), dtype=torch.int64))
        tm = self.tile_map()
        mt = self.ws.alloc(f"{self.name}.tiles", len(tm), torch.int32, zero=True)
        mt[:len(tm)].copy_(torch.tensor(tm, dtype=torch.int32))
        gp = L.GemmTNGroup()
        gp.descs, gp.tile_map, gp.n_descs, gp.n_blocks = dt.data_ptr(), mt.data_ptr(), len(self.descs), len(tm)
        gp.tile = self.tile
        if self.cursor is None:
            self.cursor = sum(1 for r in tm if r >= 0) > self.CURSOR_AUTO_TILES
        if self.tile == 128 and self.cursor:
            # row cursor (aew_gemm_tn_group_t.cursors): one progress word per tile of a descriptor, zeroed by an op right in
            # front of the launch; whether the kernel uses it is the tuning record's decision (aew_set_tn_cursor)
            gp.cursor_stride = 64
            cur = self.ws.alloc(f"{self.name}.cursors", len(self.descs) * gp.cursor_stride, torch.int32, zero=True)
            gp.cursors = cur.data_ptr()
            plan.zero(self.ws, f"{self.name}.cursors")
        # host copy of the descriptors by op label (tools/op_roofline.py prices the group from them; the launcher only
        # sees the device table)
        if not hasattr(plan, "tn_groups"):
            plan.tn_groups = {}
        plan.tn_groups[label] = list(self.descs)
        return plan.add(L.OP_GEMM_TN_GROUP, gp, label, tag, join=join)


def split_small_nt(plan: "Plan", ws: Workspace, name: str, target_blocks: int = 256) -> List[Tuple[str, int]]:
    """Split-K hint for the bf16 NT ops of `plan` that launch a few dozen 64 x 64 blocks on 256 CUs (aew_gemm_nt_t.k_split;
    aew_gemm_nt_small_split states which and how far: wavenet.py:275 / wave_encoder.py:39 backward - the upsampler and
    encoder data gradients).  Allocates each op's partial-sum slabs and tickets.  Returns [(label, S)]."""
    lib = L.load()
    out = []
    for op, lab in zip(plan.ops, plan.labels):
        if op.kind != L.OP_GEMM_NT or op.u.nt.dtype != L.BF16 or op.u.nt.impl != 0 or op.u.nt.k_split > 1:
            continue
        S, nbytes, ntk = C.c_int(1), C.c_int64(0), C.c_int(0)
        L.check(lib.aew_gemm_nt_small_split(C.byref(op.u.nt), int(target_blocks), C.byref(S), C.byref(nbytes), C.byref(ntk)), "small_split")
        if S.value < 2:
            continue
        wsb = ws.alloc(f"{name}.{lab}.ws", nbytes.value // 4, torch.float32)
        tk = ws.alloc(f"{name}.{lab}.tickets", ntk.value, torch.int32, zero=True)
        op.u.nt.k_split, op.u.nt.ksplit_ws, op.u.nt.ksplit_tickets = S.value, wsb.data_ptr(), tk.data_ptr()
        out.append((lab, S.value))
    return out


def insert_nt_chains(plan: "Plan", ws: Workspace, name: str, select, max_len: int = 0, force: bool = False,
                     spin_max: int = 0, flags: int = 0, max_stage_tiles: int = 0, sticky_ptr: int = 0,
                     tuning: Optional["L.Tuning"] = None) -> List[Tuple[int, int]]:
    """Chained NT launches (AEW_OP_NT_CHAIN, aewavenet.h): runs of consecutive main-lane bf16 NT ops of `plan` whose label
    passes `select` - no join except at the head of the run - get a chain op in front of them that launches the whole
    run as ONE kernel with tile-granular hand-off between the stages (wavenet.py:354-357: the layer loop; its backward).
    The stage ops stay in the plan: per-op timing, the CPU plan interpreter and aew_tuning_t.nt_chain = 0 execute them
    one by one, with the same result bit for bit.  max_len: stages per launch (0 = the whole run; 2 = pairs).  Runs the
    host-side builder (aew_nt_chain_build) refuses - shapes outside the default kernels, dependencies it cannot express
    - stay as they are.  The counters of all chains of one call share ONE buffer that a single zero op in front of the
    first chain clears (the chain launches then skip their own clearing: aew_nt_chain_t.flags & 2).
    max_stage_tiles > 0: runs whose stages average more 256 x 128 tiles than this stay unchained.
    sticky_ptr: device word no launch clears; a wait that gives up leaves (stage + 1) there (aew_nt_chain_t.sticky).
    tuning: the record the plan will RUN under (None: the process-wide one) - the stage table is built under it and a launch
    under a record with another one-window limit falls back to the stage ops (aew_nt_chain_t.built_window).
    Returns [(index of the chain op, stages)]."""
    lib = L.load()
    built_window = int((tuning if tuning is not None else L.current_tuning()).nt_window)
    made = []                                              # (first stage index in the ORIGINAL plan, n, stages, bmap, nb, nc, set)
    i = 0
    while i < len(plan.ops):
        op = plan.ops[i]
        if not (op.kind == L.OP_GEMM_NT and op.lane == 0 and select(plan.labels[i])):   # (main lane only: side-lane chains -
            i += 1                                                                      # DecoderPlan.split_chains - stay as they are)
            continue
        j = i + 1
        while j < len(plan.ops) and plan.ops[j].kind == L.OP_GEMM_NT and select(plan.labels[j]) and \
                plan.ops[j].lane == op.lane and plan.ops[j].join == 0 and (max_len <= 0 or j - i < max_len):
            j += 1
        n = j - i
        if n >= 2 and max_stage_tiles > 0:
            # a run whose stages are many tile waves each gains nothing from chaining (the fill / drain of a launch is a few
            # us against hundreds) and pays the hand-off: left as stand-alone launches
            tiles = sum(-(-plan.ops[q].u.nt.M // 256) * plan.ops[q].u.nt.batch * (plan.ops[q].u.nt.N_pad // 128) for q in range(i, j))
            if tiles > max_stage_tiles * n:
                n = 0
        if n >= 2:
            descs = (L.GemmNT * n)(*[plan.ops[q].u.nt for q in range(i, j)])
            stages = (L.NtStage * n)()
            cap = sum(((-(-d.M // 256) * d.batch + 7) // 8) * 8 * (d.N_pad // 128) for d in descs)
            bmap = (C.c_uint16 * (cap // 8))()
            nb, nc, st = C.c_int(0), C.c_int(0), C.c_int(0)
            rc = lib.aew_nt_chain_build_tuned(C.byref(descs), n, C.byref(stages), C.byref(bmap), cap, C.byref(nb), C.byref(nc),
                                              C.byref(st), int(force), C.byref(tuning) if tuning is not None else None)
            if rc != L.E_UNSUP:
                L.check(rc, f"aew_nt_chain_build ({plan.labels[i]} .. {plan.labels[j - 1]})")
                made.append((i, n, stages, bmap, nb.value, nc.value, st.value))
        i = j
    if not made:
        return []
    # one counter buffer: per chain its counters + 8 words (timeout flag, wait statistics), 16-byte aligned slices
    sizes = [((m[5] + 8 + 3) // 4) * 4 for m in made]
    cd = ws.alloc(f"{name}.counters", sum(sizes), torch.int32, zero=True)
    out: List[Tuple[int, int]] = []
    if not hasattr(plan, "nt_chains"):
        plan.nt_chains = {}
    shift, off = 0, 0
    z = L.Zero()
    z.ptr, z.bytes = cd.data_ptr(), 4 * sum(sizes)
    zop = L.Op()
    zop.kind, zop.tag, zop.lane, zop.join = L.OP_ZERO, 0, 0, 0
    zop.u.zero = z
    plan.ops.insert(made[0][0], zop)
    plan.labels.insert(made[0][0], f"zero:{name}.counters")
    shift += 1
    for k, (i0, n, stages, bmap, nb, nc, st) in enumerate(made):
        i = i0 + shift
        raw = bytes(stages)
        sd = ws.alloc(f"{name}.{k}.stages", (len(raw) + 7) // 8, torch.int64, zero=True)
        sd[:(len(raw) + 7) // 8].copy_(torch.frombuffer(bytearray(raw + b"\0" * (-len(raw) % 8)), dtype=torch.int64))
        md = ws.alloc(f"{name}.{k}.map", len(bmap), torch.int16, zero=True)
        md[:len(bmap)].copy_(torch.frombuffer(bytearray(bytes(bmap)), dtype=torch.int16))
        ch = L.NtChain()
        ch.stages, ch.block_stage, ch.counters = sd.data_ptr(), md.data_ptr(), cd.data_ptr() + 4 * off
        ch.n_stages, ch.n_blocks, ch.n_counters, ch.set, ch.n_ops, ch.spin_max = n, nb, nc, st, n, spin_max
        ch.flags = flags | 2
        ch.built_window = built_window
        ch.sticky = sticky_ptr or None
        first = plan.ops[i]
        cop = L.Op()
        cop.kind, cop.tag, cop.lane, cop.join = L.OP_NT_CHAIN, 0, first.lane, first.join
        cop.u.chain = ch
        plan.ops.insert(i, cop)
        plan.labels.insert(i, f"chain[{plan.labels[i]}..{plan.labels[i + n - 1]}]")
        plan.nt_chains[plan.labels[i]] = (list(stages), cd[off:off + sizes[k]])   # host copy of the stage table + its counters
        out.append((i, n))
        shift += 1
        off += sizes[k]
    plan._arr = None
    return out


def chain_stats(plan: "Plan") -> Dict[str, Tuple[int, int, int]]:
    """Per chained launch of the plan, from its last run: (timeout flag = stage + 1 of a wait that gave up or 0, tiles that
    found a producer unfinished, longest wait in polls)."""
    out = {}
    for lab, (stages, cd) in getattr(plan, "nt_chains", {}).items():
        n = sum(s.n_mt * s.g.batch for s in stages)
        v = cd[n:n + 3].cpu().tolist()
        out[lab] = (int(v[0]), int(v[1]), int(v[2]))
    return out


def chain_timeouts(plan: "Plan") -> List[str]:
    """Labels of the plan's chained launches whose last run left the timeout flag set (a hand-off wait gave up)."""
    bad = []
    for lab, (stages, cd) in getattr(plan, "nt_chains", {}).items():
        if int(cd[sum(s.n_mt * s.g.batch for s in stages)].item()) != 0:
            bad.append(lab)
    return bad


class CopyTableBuilder:
    """Collects strided-copy records and uploads them as one table op."""

    def __init__(self, ws: Workspace, name: str):
        self.ws, self.name = ws, name
        self.recs: List[L.CopyRec] = []

    def add(self, src_ptr: int, dst_ptr: int, dims: Sequence[int], ss: Sequence[int],
            ds: Sequence[int], src_dtype: int, dst_dtype: int, red_n: int = 1, red_stride: int = 0,
            accumulate: bool = False, scale: float = 1.0):
        dims, ss, ds = list(dims), list(ss), list(ds)
        while len(dims) < 4:
            dims.insert(0, 1); ss.insert(0, 0); ds.insert(0, 0)
        tiled = self._tiled_form(dims, ss, ds, src_dtype, dst_dtype, red_n, accumulate) if self.tiled else None
        if tiled is None and self.interleave:
            tiled = self._interleave_form(dims, ss, ds, src_ptr, dst_ptr, src_dtype, dst_dtype, red_n, accumulate)
        if tiled is not None:
            dims, ss, ds, tr_a, tr_b = tiled
            r = L.CopyRec()
            r.src, r.dst = src_ptr, dst_ptr
            for i in range(4):
                r.dims[i], r.ss[i], r.ds[i] = dims[i], ss[i], ds[i]
            r.src_dtype, r.dst_dtype, r.red_n, r.red_stride = src_dtype, dst_dtype, 1, 0
            r.accumulate, r.scale, r.tr_a, r.tr_b = 0, scale, tr_a, tr_b
            self.recs.append(r)
            return
        # thread index runs fastest over the LAST dim: order dims so that it is the one with the
        # smallest source stride (coalesced reads; the read side carries the slab reduction)
        order = sorted(range(4), key=lambda i: (0, 0) if dims[i] == 1 else (1, -abs(ss[i])))
        if src_dtype == L.F32 and dst_dtype == L.F32 and red_n == 1 and not accumulate:
            # permuting fp32 copies (encoder weight pack): destination-contiguous dim last unless the
            # source-contiguous one already is (then the float4-load form applies)
            cand = [i for i in range(4) if ds[i] == 1 and dims[i] % 4 == 0 and dims[i] > 1]
            if cand and ds[order[-1]] != 1:
                order = [i for i in order if i != cand[0]] + [cand[0]]
        if src_dtype == L.F32 and dst_dtype == L.BF16 and red_n == 1:
            # fp32 -> bf16 packs: the kernel's destination-vector form wants the dim that is contiguous in
            # the DESTINATION last (8 strided loads, one 16-byte store)
            cand = [i for i in range(4) if ds[i] == 1 and dims[i] % 8 == 0 and dims[i] > 1]
            if cand and ds[order[-1]] != 1:
                order = [i for i in order if i != cand[0]] + [cand[0]]
        dims, ss, ds = [dims[i] for i in order], [ss[i] for i in order], [ds[i] for i in order]
        r = L.CopyRec()
        r.src, r.dst = src_ptr, dst_ptr
        for i in range(4):
            r.dims[i], r.ss[i], r.ds[i] = dims[i], ss[i], ds[i]
        r.src_dtype, r.dst_dtype, r.red_n, r.red_stride = src_dtype, dst_dtype, red_n, red_stride
        r.accumulate, r.scale = int(accumulate), scale
        self.recs.append(r)

    tiled = True              # transposing records go through LDS tiles (False: element-wise forms only; A/B)

    @staticmethod
    def _tiled_form(dims, ss, ds, src_dtype, dst_dtype, red_n, accumulate):
        """A record that is contiguous along one dim in the source and along ANOTHER in the destination (transposing
        weight packs, tap <-> channel permutations) -> (dims, ss, ds, tr_a, tr_b) with dims[3] the source-side and
        dims[2] the destination-side dim and the tile the kernel moves through LDS, or None when the element-wise
        forms already run along both (k_copy_table)."""
        if src_dtype != L.F32 or dst_dtype not in (L.F32, L.BF16) or red_n != 1 or accumulate:
            return None
        dims, ss, ds = list(dims), list(ss), list(ds)
        # merge dims that are nested the same way on both sides ((c, tap) of a conv weight is one run of c * k elements)
        merged = True
        while merged:
            merged = False
            for i in range(4):
                for j in range(4):
                    if i != j and dims[i] > 1 and dims[j] > 1 and ss[i] == ss[j] * dims[j] and ds[i] == ds[j] * dims[j]:
                        dims[j] *= dims[i]
                        dims[i], ss[i], ds[i] = 1, 0, 0
                        merged = True
        live = [i for i in range(4) if dims[i] > 1]
        bs = [i for i in live if ds[i] == 1]
        if not bs:
            return None
        b = bs[0]
        rest = [i for i in live if i != b]
        if not rest:
            return None
        a = min(rest, key=lambda i: abs(ss[i]))
        A, B = dims[a], dims[b]
        if not (0 < ss[a] <= 2):
            return None

        def fit(extent, cap):
            """Tile extent (a power of two <= cap, >= 16) along a dim: the largest unless a smaller one covers the dim
            with >= 5 % less padding."""
            best = None
            for t in (cap, cap // 2, cap // 4):
                if t < 16:
                    continue
                util = extent / (-(-extent // t) * t)
                if best is None or util > best[0] + 0.05:
                    best = (util, t)
            return best[1]

        def pow2_le(n):
            return 1 << (n.bit_length() - 1)
        if ss[b] > 8 and ds[a] > 8 and B >= 32:
            # a transpose proper: the element-wise forms run along ONE side (8 loads at stride ss[b] per 16-byte store, or
            # one 16-byte load per 4 stores at stride ds[a]) and touch a cache line per element on the other.  Measured
            # slower tiled and left element-wise: B < 32 (the gate-permuted 16-wide runs: decoder pack 52 -> 67 us) and
            # the tap <-> channel (de)interleaves inside a row, whose short strides the element-wise forms already
            # cover (k x 256 tiles: decoder pack / unpack 59 -> 85 / 49 -> 76 us).
            cap = CopyTableBuilder.tile_cap
            ta = A if A <= 8 else (16 if A <= 16 else (32 if (A <= 32 or cap < 4096) else 64))
            tb = fit(B, pow2_le(min(cap // ta, (4160 if cap >= 4096 else 1600) // (ta | 1))))     # (LDS pitch ta | 1)
        else:
            return None
        assert ta * tb <= 4096 and tb * (ta | 1) <= 4160
        outer = [i for i in range(4) if i not in (a, b)]
        outer.sort(key=lambda i: dims[i] > 1)                      # size-1 dims first
        order = outer + [b, a]
        return [dims[i] for i in order], [ss[i] for i in order], [ds[i] for i in order], ta, tb

    tile_cap = 4096           # elements per LDS tile of a transposing record (1024: the round-3 tiles, 4 loads per thread in flight)
    interleave = True         # tap <-> channel (de)interleaves as register permutations (False: element-wise forms; A/B)

    @staticmethod
    def _interleave_form(dims, ss, ds, src_ptr, dst_ptr, src_dtype, dst_dtype, red_n, accumulate):
        """Conv-weight records that swap the tap dim (k = 2..4) with the channel dim inside a row:
        [..][k][c] -> [..][c][k] (gradient unpack, tr_b = 1) and [..][c][k] -> [..][k][c] (weight pack, tr_b = 2).
        -> (dims, ss, ds, -k, tr_b) in the kernel's dim order, or None (shape or alignment not covered)."""
        if src_dtype != L.F32 or dst_dtype not in (L.F32, L.BF16) or red_n != 1 or accumulate:
            return None
        dims, ss, ds = list(dims), list(ss), list(ds)
        live = [i for i in range(4) if dims[i] > 1]
        a = [i for i in live if ss[i] == 1]
        b = [i for i in live if ds[i] == 1]
        if len(a) != 1 or len(b) != 1 or a[0] == b[0]:
            return None
        a, b = a[0], b[0]
        rest = [i for i in range(4) if i not in (a, b)]
        rest.sort(key=lambda i: dims[i] > 1)
        if dims[b] in (2, 3, 4) and ds[a] == dims[b] and dst_dtype == L.F32:
            k, mode = dims[b], 1                                    # interleave: a = channels (source run), b = taps
            ok = dims[a] % 4 == 0 and ss[b] % 4 == 0 and all(ss[i] % 4 == 0 and ds[i] % 4 == 0 for i in rest if dims[i] > 1)
            order = rest + [b, a]
        elif dims[a] in (2, 3, 4) and ss[b] == dims[a]:
            k, mode = dims[a], 2                                    # de-interleave: a = taps, b = channels (destination run)
            w = 8 if dst_dtype == L.BF16 else 4
            ok = dims[b] % w == 0 and ds[a] % w == 0 and all(ss[i] % 4 == 0 and ds[i] % w == 0 for i in rest if dims[i] > 1)
            order = rest + [b, a]
        else:
            return None
        if not ok or src_ptr % 16 or dst_ptr % 16:
            return None
        return [dims[i] for i in order], [ss[i] for i in order], [ds[i] for i in order], -k, mode

    def emit(self, plan: Plan, label: str, join: bool = False):
        if not self.recs:
            return
        block_rec: List[int] = []
        for i, r in enumerate(self.recs):
            n = r.dims[0] * r.dims[1] * r.dims[2] * r.dims[3]
            r.first_block = len(block_rec)
            if r.tr_a > 0:
                block_rec.extend([i] * (r.dims[0] * r.dims[1] * (-(-r.dims[2] // r.tr_b)) * (-(-r.dims[3] // r.tr_a))))
                continue
            if r.tr_a < 0:                                         # interleave forms: one item per W channels of all taps
                w = 8 if (r.tr_b == 2 and r.dst_dtype == L.BF16) else 4
                items = r.dims[0] * r.dims[1] * ((r.dims[3] if r.tr_b == 1 else r.dims[2]) // w)
                block_rec.extend([i] * ((items + 255) // 256))
                continue
            block_rec.extend([i] * ((n + 1023) // 1024))
        raw = bytes((L.CopyRec * len(self.recs))(*self.recs))
        rec_t = self.ws.alloc(f"{self.name}.recs", (len(raw) + 7) // 8, torch.int64, zero=True)
        host = torch.frombuffer(bytearray(raw + b"\0" * (-len(raw) % 8)), dtype=torch.int64)
        rec_t[:host.numel()].copy_(host)
        blk_t = self.ws.alloc(f"{self.name}.blocks", len(block_rec), torch.int32, zero=True)
        blk_t[:len(block_rec)].copy_(torch.tensor(block_rec, dtype=torch.int32))
        t = L.CopyTable()
        t.recs, t.block_rec = rec_t.data_ptr(), blk_t.data_ptr()
        t.n_blocks, t.n_recs = len(block_rec), len(self.recs)
        plan.add(L.OP_COPY_TABLE, t, label, join=join)
