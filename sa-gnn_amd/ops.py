"""Tensor-level wrappers over the C ABI (include/sagnn.h): PyTorch-ROCm tensors in, HIP kernels
out. torch is used for device memory and streams only — no arithmetic happens in torch here."""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import PlanInfo, Tuning, check

SUPPORTED_D = tuple(range(4, 257, 4))

# sagnn_set_engine (include/sagnn.h): arithmetic engine of the GEMM-shaped fusion stages, per calling thread
ENGINES = {"f16x2": 0, "f32": 1, "valu": 2}


def set_engine(name: str) -> None:
    check(_lib.load().sagnn_set_engine(ENGINES[name]))


def get_engine() -> str:
    code = _lib.load().sagnn_get_engine()
    return next(k for k, v in ENGINES.items() if v == code)


def range_redo_count(reset: bool = False) -> int:
    """sagnn_range_redo_count: tiles / chunks the f16 x 2 kernels re-evaluated in fp32 since the last reset."""
    n = ctypes.c_int64(0)
    check(_lib.load().sagnn_range_redo_count(ctypes.byref(n), 1 if reset else 0))
    return int(n.value)


class engine:
    """`with ops.engine("f32"): ...` — runs the block under that engine and restores the caller's."""

    def __init__(self, name: str):
        self.name = name

    def __enter__(self):
        self.prev = get_engine()
        set_engine(self.name)
        return self

    def __exit__(self, *exc):
        set_engine(self.prev)
        return False


def _stream() -> int:
    # the raw handle of torch's current stream; torch.cuda.current_stream() builds a Stream object per call
    # (10 us, a few hundred times per training step)
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _ptr(t: torch.Tensor | None) -> int | None:
    return None if t is None else t.data_ptr()


def _f32_rows(name: str, t: torch.Tensor | None, d: int, rows: int | None = None):
    """Checks a [rows, d] fp32 device matrix view with unit inner stride; returns its row stride."""
    if t is None:
        return 0
    if t.dtype != torch.float32 or not t.is_cuda:
        raise TypeError(f"{name}: expected a float32 device tensor, got {t.dtype} on {t.device}")
    if t.dim() != 2 or t.shape[1] != d or t.stride(1) != 1:
        raise ValueError(f"{name}: expected shape [rows, {d}] with unit inner stride, got "
                         f"{tuple(t.shape)} strides {t.stride()}")
    if rows is not None and t.shape[0] != rows:
        raise ValueError(f"{name}: expected {rows} rows, got {t.shape[0]}")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), d)


class SpmmPlan:
    """Degree-class plan for one CSR adjacency (sagnn_spmm_plan_*). Owns the device CSR copies.

    rowptr / colidx: int32, numpy or torch (host or device). With device=None a host-only plan
    is built (no GPU touched) for inspecting the chunking."""

    def __init__(self, rowptr, colidx, n_rows: int, n_src: int, device=None,
                 tuning: tuple[int, int, int] | None = None, validate: bool = True):
        lib = _lib.load()
        self._lib = lib
        self._h = ctypes.c_void_p()
        self.n_rows, self.n_src = int(n_rows), int(n_src)
        rp_t = torch.as_tensor(rowptr)
        ci_t = torch.as_tensor(colidx)
        if rp_t.dtype != torch.int32 or ci_t.dtype != torch.int32:
            raise TypeError("rowptr/colidx must be int32")
        if rp_t.numel() != self.n_rows + 1:
            raise ValueError(f"rowptr has {rp_t.numel()} entries, expected n_rows+1 = {self.n_rows + 1}")
        self.nnz = int(ci_t.numel())
        rp_h = rp_t.cpu().contiguous()
        self._rowptr_host = rp_h.numpy()
        if validate:
            ci_h = ci_t.cpu().contiguous()
            check(lib.sagnn_csr_check_host(rp_h.data_ptr(), ci_h.data_ptr(), self.n_rows, self.n_src,
                                           self.nnz))
        tun = None
        if tuning is not None:
            tun = Tuning(int(tuning[0]), int(tuning[1]), int(tuning[2]), 0)
        if device is None:
            self.rowptr = self.colidx = None
            d_rp = d_ci = None
        else:
            self.rowptr = rp_t.to(device).contiguous()
            self.colidx = ci_t.to(device).contiguous()
            if self.colidx.numel() == 0:  # keep a valid pointer for the (unused) argument
                self.colidx = torch.zeros(1, dtype=torch.int32, device=device)
            d_rp, d_ci = self.rowptr.data_ptr(), self.colidx.data_ptr()
        check(lib.sagnn_spmm_plan_create(rp_h.data_ptr(), d_rp, d_ci, self.n_rows, self.n_src,
                                         self.nnz, ctypes.byref(tun) if tun else None,
                                         ctypes.byref(self._h)))
        info = PlanInfo()
        check(lib.sagnn_spmm_plan_get_info(self._h, ctypes.byref(info)))
        self.info = info
        self.device = device
        self._ws: dict[int, torch.Tensor] = {}
        # plan of the partner's exact transpose when the (user, item) pair of an interval is not a
        # transposed pair (duplicated stored entries, graph.interval_pair); None = the pair is exact
        self.partner_adjoint: SpmmPlan | None = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.sagnn_spmm_plan_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def chunks(self):
        """(rows, e_begin, e_end) int32 arrays of the long-row chunk list."""
        n = int(self.info.n_chunks)
        rows = np.empty(n, np.int32)
        e0 = np.empty(n, np.int32)
        e1 = np.empty(n, np.int32)
        check(self._lib.sagnn_spmm_plan_copy_chunks(self._h, rows.ctypes.data, e0.ctypes.data,
                                                    e1.ctypes.data, n))
        return rows, e0, e1

    def workspace_bytes(self, d: int) -> int:
        return int(self._lib.sagnn_spmm_workspace_bytes(self._h, int(d)))

    def workspace(self, d: int) -> torch.Tensor | None:
        need = self.workspace_bytes(d)
        if need == 0:
            return None
        ws = self._ws.get(d)
        if ws is None:
            ws = torch.empty(need // 4, dtype=torch.float32, device=self.device)
            self._ws[d] = ws
        return ws


def spmm(plan: SpmmPlan, x: torch.Tensor, leaky: float, residual: torch.Tensor | None = None,
         out: torch.Tensor | None = None, acc_in: torch.Tensor | None = None,
         acc_out: torch.Tensor | None = None, want_out: bool = True) -> torch.Tensor | None:
    """y = max(leaky*(A·x), A·x) + residual;  out = y;  acc_out = acc_in + y  (sagnn_spmm_f32).

    Replaces Recommender.messagePropagate (reference model.py:80-92) and, through
    residual/acc_*, the adds of model.py:124-127. Returns `out` (allocated if want_out and not
    given)."""
    d = int(x.shape[1])
    ldx = _f32_rows("x", x, d, plan.n_src)
    if out is None and want_out:
        out = torch.empty((plan.n_rows, d), dtype=torch.float32, device=x.device)
    ldr = _f32_rows("residual", residual, d, plan.n_rows)
    ldo = _f32_rows("out", out, d, plan.n_rows)
    ldai = _f32_rows("acc_in", acc_in, d, plan.n_rows)
    ldao = _f32_rows("acc_out", acc_out, d, plan.n_rows)
    ws = plan.workspace(d)
    check(plan._lib.sagnn_spmm_f32(plan.handle, _ptr(x), ldx, d, _ptr(residual), ldr, float(leaky),
                                   _ptr(out), ldo, _ptr(acc_in), ldai, _ptr(acc_out), ldao,
                                   _ptr(ws), 0 if ws is None else ws.numel() * 4, _stream()))
    return out


def spmm_ex(plan: SpmmPlan, x: torch.Tensor | None, leaky: float, residual=None, out=None, acc_in=None, acc_out=None,
            acc_in2=None, mask_out=None, mask_in=None, out2=None, slope2: float = 1.0, want_out: bool = False):
    """sagnn_spmm_ex_f32: spmm plus the training epilogue — mask_out [rows, d/4] uint8 records the activation slopes,
    out2 = v * (mask_in bit ? 1 : slope2) with v the accumulated value if acc_out is given, acc_in2 a second addend."""
    ref = next(t_ for t_ in (x, residual, out, acc_out, out2) if t_ is not None)
    d = int(ref.shape[1])
    if out is None and want_out:
        out = torch.empty((plan.n_rows, d), dtype=torch.float32, device=ref.device)
    e = _lib.SpmmEpilogue()
    e.leaky, e.slope2 = float(leaky), float(slope2)
    e.residual, e.ldr = _ptr(residual), _f32_rows("residual", residual, d, plan.n_rows)
    e.out, e.ldo = _ptr(out), _f32_rows("out", out, d, plan.n_rows)
    e.acc_in, e.ld_acc_in = _ptr(acc_in), _f32_rows("acc_in", acc_in, d, plan.n_rows)
    e.acc_out, e.ld_acc_out = _ptr(acc_out), _f32_rows("acc_out", acc_out, d, plan.n_rows)
    e.acc_in2, e.ld_acc_in2 = _ptr(acc_in2), _f32_rows("acc_in2", acc_in2, d, plan.n_rows)
    e.out2, e.ldo2 = _ptr(out2), _f32_rows("out2", out2, d, plan.n_rows)
    for name, m in (("mask_out", mask_out), ("mask_in", mask_in)):
        if m is not None and (m.dtype != torch.uint8 or not m.is_contiguous() or m.numel() != plan.n_rows * (d // 4)):
            raise ValueError(f"{name}: need a contiguous uint8 tensor [{plan.n_rows}, {d // 4}]")
    e.mask_out, e.mask_in = _ptr(mask_out), _ptr(mask_in)
    ldx = _f32_rows("x", x, d, plan.n_src) if x is not None else d
    ws = plan.workspace(d)
    check(plan._lib.sagnn_spmm_ex_f32(plan.handle, _ptr(x), ldx, d, ctypes.byref(e), _ptr(ws), 0 if ws is None else ws.numel() * 4,
                                      _stream()))
    return out


def mask_scale(g: torch.Tensor, mask: torch.Tensor, slope: float, out: torch.Tensor):
    """out = g * (mask bit ? 1 : slope) (sagnn_mask_scale_f32); g / out [rows, d] views, mask [rows, d/4] uint8."""
    rows, d = int(g.shape[0]), int(g.shape[1])
    if mask.dtype != torch.uint8 or not mask.is_contiguous() or mask.numel() != rows * (d // 4):
        raise ValueError(f"mask: need a contiguous uint8 tensor [{rows}, {d // 4}]")
    check(_lib.load().sagnn_mask_scale_f32(_ptr(g), _f32_rows("g", g, d, rows), _ptr(mask), float(slope), _ptr(out),
                                           _f32_rows("out", out, d, rows), rows, d, _stream()))
    return out


def _interval_ws(plan_user: SpmmPlan, plan_item: SpmmPlan, d: int):
    wu, wi = plan_user.workspace(d), plan_item.workspace(d)
    return wu if (wi is None or (wu is not None and wu.numel() >= wi.numel())) else wi


def gnn_interval(plan_user: SpmmPlan, plan_item: SpmmPlan, u0: torch.Tensor, i0: torch.Tensor,
                 n_layers: int, leaky: float, user_out: torch.Tensor, item_out: torch.Tensor,
                 scratch_u: torch.Tensor | None = None, scratch_i: torch.Tensor | None = None,
                 mask_u: torch.Tensor | None = None, mask_i: torch.Tensor | None = None):
    """One interval of the GNN loop (reference model.py:118-129): sagnn_gnn_interval_[ex_]f32.
    user_out / item_out are [rows, d] views (any row stride, e.g. a column of an [N, T, d] slab).
    mask_u [L, U, d/4] / mask_i [L, I, d/4] uint8 (both or neither) record the activation masks
    the backward pass needs."""
    d = int(u0.shape[1])
    U, I = plan_user.n_rows, plan_item.n_rows
    ld_u0 = _f32_rows("u0", u0, d, U)
    ld_i0 = _f32_rows("i0", i0, d, I)
    ld_uo = _f32_rows("user_out", user_out, d, U)
    ld_io = _f32_rows("item_out", item_out, d, I)
    if n_layers > 1:
        if scratch_u is None:
            scratch_u = torch.empty((2, U, d), dtype=torch.float32, device=u0.device)
        if scratch_i is None:
            scratch_i = torch.empty((2, I, d), dtype=torch.float32, device=u0.device)
        for name, s, rows in (("scratch_u", scratch_u, U), ("scratch_i", scratch_i, I)):
            if s.dtype != torch.float32 or not s.is_contiguous() or s.numel() < 2 * rows * d:
                raise ValueError(f"{name}: need a contiguous float32 buffer of 2*{rows}*{d} elements")
    for name, m, rows in (("mask_u", mask_u, U), ("mask_i", mask_i, I)):
        if m is not None and (m.dtype != torch.uint8 or not m.is_contiguous() or
                              m.numel() != n_layers * rows * (d // 4)):
            raise ValueError(f"{name}: need a contiguous uint8 tensor [{n_layers}, {rows}, {d // 4}]")
    ws = _interval_ws(plan_user, plan_item, d)
    check(plan_user._lib.sagnn_gnn_interval_ex_f32(
        plan_user.handle, plan_item.handle, _ptr(u0), ld_u0, _ptr(i0), ld_i0, d, int(n_layers),
        float(leaky), _ptr(scratch_u), _ptr(scratch_i), _ptr(user_out), ld_uo, _ptr(item_out), ld_io,
        _ptr(mask_u), _ptr(mask_i), _ptr(ws), 0 if ws is None else ws.numel() * 4, _stream()))
    return user_out, item_out


def gnn_interval_bwd(plan_user: SpmmPlan, plan_item: SpmmPlan, grad_user_out: torch.Tensor,
                     grad_item_out: torch.Tensor, n_layers: int, leaky: float, mask_u: torch.Tensor,
                     mask_i: torch.Tensor, grad_u0: torch.Tensor | None = None,
                     grad_i0: torch.Tensor | None = None, scratch_u: torch.Tensor | None = None,
                     scratch_i: torch.Tensor | None = None):
    """Backward of gnn_interval (sagnn_gnn_interval_bwd_f32): dL/d(user_out), dL/d(item_out) and the
    recorded masks -> dL/d u0 [U, d], dL/d i0 [I, d]."""
    d = int(grad_user_out.shape[1])
    U, I = plan_user.n_rows, plan_item.n_rows
    # rows = users gathers through (item-side forward pattern)^T, rows = items through (user-side)^T
    adj_u, adj_i = plan_user.partner_adjoint, plan_item.partner_adjoint
    if (adj_u is None) != (adj_i is None):
        raise ValueError("give the exact adjoint of both plans or of neither")
    if adj_u is None and plan_user.nnz != plan_item.nnz:
        raise ValueError(f"plans are not a transposed pair (nnz {plan_user.nnz} vs {plan_item.nnz}: duplicated stored "
                         "entries?) — build them with graph.interval_pair, which adds the exact adjoints")
    if adj_u is not None:
        plan_user, plan_item = adj_u, adj_i
    ld_gu = _f32_rows("grad_user_out", grad_user_out, d, U)
    ld_gi = _f32_rows("grad_item_out", grad_item_out, d, I)
    dev = grad_user_out.device
    if grad_u0 is None:
        grad_u0 = torch.empty((U, d), dtype=torch.float32, device=dev)
    if grad_i0 is None:
        grad_i0 = torch.empty((I, d), dtype=torch.float32, device=dev)
    ld_du = _f32_rows("grad_u0", grad_u0, d, U)
    ld_di = _f32_rows("grad_i0", grad_i0, d, I)
    if scratch_u is None:
        scratch_u = torch.empty((4, U, d), dtype=torch.float32, device=dev)
    if scratch_i is None:
        scratch_i = torch.empty((4, I, d), dtype=torch.float32, device=dev)
    for name, s_, rows in (("scratch_u", scratch_u, U), ("scratch_i", scratch_i, I)):
        if s_.dtype != torch.float32 or not s_.is_contiguous() or s_.numel() < 4 * rows * d:
            raise ValueError(f"{name}: need a contiguous float32 buffer of 4*{rows}*{d} elements")
    for name, m, rows in (("mask_u", mask_u, U), ("mask_i", mask_i, I)):
        if m.dtype != torch.uint8 or not m.is_contiguous() or m.numel() != n_layers * rows * (d // 4):
            raise ValueError(f"{name}: need a contiguous uint8 tensor [{n_layers}, {rows}, {d // 4}]")
    ws = _interval_ws(plan_user, plan_item, d)
    check(plan_user._lib.sagnn_gnn_interval_bwd_f32(
        plan_user.handle, plan_item.handle, _ptr(grad_user_out), ld_gu, _ptr(grad_item_out), ld_gi, d,
        int(n_layers), float(leaky), _ptr(mask_u), _ptr(mask_i), _ptr(scratch_u), _ptr(scratch_i),
        _ptr(grad_u0), ld_du, _ptr(grad_i0), ld_di, _ptr(ws), 0 if ws is None else ws.numel() * 4, _stream()))
    return grad_u0, grad_i0


class SpmmBatch:
    """The T interval plans of a model tied into one launch per layer (sagnn_spmm_batch_*): what the reference's
    `for k in range(args.graphNum)` loop (model.py:118) becomes when its 2 T L SpMMs are launch-bound. Keeps the
    plans alive. `adjoint()` is the batch the backward pass runs on (the same one unless a matrix holds duplicated
    stored entries, graph.interval_pair)."""

    def __init__(self, plans_user, plans_item):
        if len(plans_user) != len(plans_item) or not plans_user:
            raise ValueError("one user-side and one item-side plan per interval")
        self.plans_user, self.plans_item = list(plans_user), list(plans_item)
        self.T, self.U, self.I = len(plans_user), plans_user[0].n_rows, plans_item[0].n_rows
        self.device = plans_user[0].device
        self._lib = _lib.load()
        self._h = ctypes.c_void_p()
        PU = (ctypes.c_void_p * self.T)(*[p.handle for p in self.plans_user])
        PI = (ctypes.c_void_p * self.T)(*[p.handle for p in self.plans_item])
        check(self._lib.sagnn_spmm_batch_create(PU, PI, self.T, ctypes.byref(self._h)))
        self.nnz = sum(p.nnz for p in self.plans_user) + sum(p.nnz for p in self.plans_item)
        self._ws: dict[int, torch.Tensor] = {}
        self._adjoint: SpmmBatch | None = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.sagnn_spmm_batch_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def workspace(self, d: int):
        need = int(self._lib.sagnn_spmm_batch_workspace_bytes(self._h, int(d)))
        if need == 0:
            return None
        ws = self._ws.get(d)
        if ws is None:
            ws = torch.empty(need // 4, dtype=torch.float32, device=self.device)
            self._ws[d] = ws
        return ws

    def adjoint(self) -> "SpmmBatch":
        adj_u = [p.partner_adjoint for p in self.plans_user]
        adj_i = [p.partner_adjoint for p in self.plans_item]
        if all(a is None for a in adj_u + adj_i):
            for pu, pi in zip(self.plans_user, self.plans_item):
                if pu.nnz != pi.nnz:
                    raise ValueError("plans are not a transposed pair (duplicated stored entries?) — build them with "
                                     "graph.interval_pair, which adds the exact adjoints")
            return self
        if self._adjoint is None:
            self._adjoint = SpmmBatch([a if a is not None else p for a, p in zip(adj_u, self.plans_user)],
                                      [a if a is not None else p for a, p in zip(adj_i, self.plans_item)])
        return self._adjoint


def _slab(name: str, x: torch.Tensor, T: int, rows: int, d: int):
    """x indexed [interval, row, feature] with unit feature stride (any other strides: [T, N, d] storage or a
    permuted view of [N, T, d]); returns (ld, slab) in elements."""
    if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 3 or tuple(x.shape) != (T, rows, d) or x.stride(2) != 1:
        raise ValueError(f"{name}: expected a float32 device tensor indexed [{T}, {rows}, {d}] with unit feature stride, "
                         f"got {tuple(x.shape)} strides {x.stride()}")
    ld = int(x.stride(1)) if rows > 1 else max(int(x.stride(1)), d)
    slab = int(x.stride(0)) if T > 1 else 0
    return ld, slab


def gnn_stack(batch: SpmmBatch, u0: torch.Tensor, i0: torch.Tensor, n_layers: int, leaky: float,
              user_out: torch.Tensor, item_out: torch.Tensor, scratch_u: torch.Tensor | None = None,
              scratch_i: torch.Tensor | None = None, mask_u: torch.Tensor | None = None, mask_i: torch.Tensor | None = None):
    """Every interval of the GNN loop (reference model.py:118-129) in one launch per layer: sagnn_gnn_stack_f32.
    u0 [T, U, d], i0 [T, I, d]; user_out / item_out indexed [T, N, d] (e.g. `x.permute(1, 0, 2)` of the [N, T, d]
    tensor the fusion reads); mask_u [T, L, U, d/4] / mask_i [T, L, I, d/4] uint8 for training."""
    T, U, I = batch.T, batch.U, batch.I
    d = int(u0.shape[2])
    ld_u0, sl_u0 = _slab("u0", u0, T, U, d)
    ld_i0, sl_i0 = _slab("i0", i0, T, I, d)
    ld_uo, sl_uo = _slab("user_out", user_out, T, U, d)
    ld_io, sl_io = _slab("item_out", item_out, T, I, d)
    if n_layers > 1:
        if scratch_u is None:
            scratch_u = torch.empty((2, T, U, d), dtype=torch.float32, device=u0.device)
        if scratch_i is None:
            scratch_i = torch.empty((2, T, I, d), dtype=torch.float32, device=u0.device)
        for name, s_, rows in (("scratch_u", scratch_u, U), ("scratch_i", scratch_i, I)):
            if s_.dtype != torch.float32 or not s_.is_contiguous() or s_.numel() < 2 * T * rows * d:
                raise ValueError(f"{name}: need a contiguous float32 buffer of 2*{T}*{rows}*{d} elements")
    for name, m, rows in (("mask_u", mask_u, U), ("mask_i", mask_i, I)):
        if m is not None and (m.dtype != torch.uint8 or not m.is_contiguous() or m.numel() != T * n_layers * rows * (d // 4)):
            raise ValueError(f"{name}: need a contiguous uint8 tensor [{T}, {n_layers}, {rows}, {d // 4}]")
    ws = batch.workspace(d)
    check(batch._lib.sagnn_gnn_stack_f32(batch.handle, _ptr(u0), ld_u0, sl_u0, _ptr(i0), ld_i0, sl_i0, d, int(n_layers),
                                         float(leaky), _ptr(scratch_u), _ptr(scratch_i), _ptr(user_out), ld_uo, sl_uo,
                                         _ptr(item_out), ld_io, sl_io, _ptr(mask_u), _ptr(mask_i), _ptr(ws),
                                         0 if ws is None else ws.numel() * 4, _stream()))
    return user_out, item_out


def gnn_stack_bwd(batch: SpmmBatch, grad_user_out: torch.Tensor, grad_item_out: torch.Tensor, n_layers: int, leaky: float,
                  mask_u: torch.Tensor, mask_i: torch.Tensor, grad_u0: torch.Tensor, grad_i0: torch.Tensor,
                  scratch_u: torch.Tensor | None = None, scratch_i: torch.Tensor | None = None):
    """Backward of gnn_stack (sagnn_gnn_stack_bwd_f32) on the batch's adjoint patterns: gradients at the interval
    outputs [T, N, d] (any strides) -> dL/d u0 [T, U, d], dL/d i0 [T, I, d]."""
    adj = batch.adjoint()
    T, U, I = batch.T, batch.U, batch.I
    d = int(grad_user_out.shape[2])
    ld_gu, sl_gu = _slab("grad_user_out", grad_user_out, T, U, d)
    ld_gi, sl_gi = _slab("grad_item_out", grad_item_out, T, I, d)
    ld_du, sl_du = _slab("grad_u0", grad_u0, T, U, d)
    ld_di, sl_di = _slab("grad_i0", grad_i0, T, I, d)
    dev = grad_user_out.device
    if scratch_u is None:
        scratch_u = torch.empty((4, T, U, d), dtype=torch.float32, device=dev)
    if scratch_i is None:
        scratch_i = torch.empty((4, T, I, d), dtype=torch.float32, device=dev)
    for name, m, rows in (("mask_u", mask_u, U), ("mask_i", mask_i, I)):
        if m.dtype != torch.uint8 or not m.is_contiguous() or m.numel() != T * n_layers * rows * (d // 4):
            raise ValueError(f"{name}: need a contiguous uint8 tensor [{T}, {n_layers}, {rows}, {d // 4}]")
    ws = adj.workspace(d)
    check(adj._lib.sagnn_gnn_stack_bwd_f32(adj.handle, _ptr(grad_user_out), ld_gu, sl_gu, _ptr(grad_item_out), ld_gi, sl_gi, d,
                                           int(n_layers), float(leaky), _ptr(mask_u), _ptr(mask_i), _ptr(scratch_u),
                                           _ptr(scratch_i), _ptr(grad_u0), ld_du, sl_du, _ptr(grad_i0), ld_di, sl_di, _ptr(ws),
                                           0 if ws is None else ws.numel() * 4, _stream()))
    return grad_u0, grad_i0


def _ntd(name: str, x: torch.Tensor, dense_td: bool = False):
    """x is indexed [node, interval, feature]; any node/interval strides (so a permuted view of
    [t, n, d] storage works). Returns n, t, d, ld_n, ld_t."""
    if x.dtype != torch.float32 or not x.is_cuda or x.dim() != 3:
        raise TypeError(f"{name}: expected a float32 device tensor indexed [n, t, d]")
    n, t, d = (int(v) for v in x.shape)
    if x.stride(2) != 1:
        raise ValueError(f"{name}: the feature axis must have unit stride")
    ld_n = int(x.stride(0)) if n > 1 else max(int(x.stride(0)), d)
    ld_t = int(x.stride(1)) if t > 1 else max(int(x.stride(1)), d)
    if dense_td and ld_t != d:
        raise ValueError(f"{name}: the (t, d) block of each node must be contiguous")
    return n, t, d, ld_n, ld_t


def _vec(name: str, v: torch.Tensor, numel: int):
    if v.dtype != torch.float32 or not v.is_cuda or not v.is_contiguous() or v.numel() != numel:
        raise ValueError(f"{name}: expected a contiguous float32 device tensor of {numel} elements")
    return v.data_ptr()


def _wide(d: int) -> bool:
    """d handled by the 'wide' MFMA composition (multiples of 32 other than the fused 32 / 64)."""
    return d % 32 == 0 and d not in (32, 64) and _lib.load().sagnn_get_engine() != ENGINES["valu"]


def lstm_fwd(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor, forget_bias: float = 1.0,
             drop_scale: torch.Tensor | None = None, out: torch.Tensor | None = None,
             h0: torch.Tensor | None = None, c0: torch.Tensor | None = None, c_out: torch.Tensor | None = None):
    """BasicLSTMCell over T (reference model.py:135-146): sagnn_lstm_fwd_state_f32. x [n, t, d].
    h0 [n, d] (any row stride) + c0 [n, d]: state to continue from (default: zero state, as the
    reference); c_out [n, d]: receives the cell state after the last step. Cutting a sequence into
    consecutive calls gives bit-identical results to one call."""
    n, t, d, ld, ldt = _ntd("x", x)
    if out is None:
        out = torch.empty((n, t, d), dtype=torch.float32, device=x.device)
    _, _, _, ldh, _ = _ntd("out", out, dense_td=True)
    if drop_scale is not None and (not drop_scale.is_contiguous() or drop_scale.shape != x.shape):
        raise ValueError("drop_scale must be contiguous [n, t, d]")
    if (h0 is None) != (c0 is None):
        raise ValueError("give both h0 and c0 or neither")
    ld_hi = 0
    if h0 is not None:
        ld_hi = _f32_rows("h0", h0, d, n)
        if _f32_rows("c0", c0, d, n) != d:
            raise ValueError("c0 must be contiguous [n, d]")
    if c_out is not None and _f32_rows("c_out", c_out, d, n) != d:
        raise ValueError("c_out must be contiguous [n, d]")
    lib = _lib.load()
    check(lib.sagnn_lstm_fwd_state_f32(x.data_ptr(), ld, ldt, n, t, d, _vec("W", W, 8 * d * d),
                                       _vec("b", b, 4 * d), float(forget_bias), _ptr(drop_scale), _ptr(h0), ld_hi,
                                       _ptr(c0), out.data_ptr(), ldh, _ptr(c_out), _stream()))
    return out


def layernorm_td(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-12,
                 out: torch.Tensor | None = None):
    """layer_norm over (t, d) per node (reference model.py:152-153): sagnn_layernorm_td_f32."""
    n, t, d, ld, ldt = _ntd("x", x)
    if out is None:
        out = torch.empty((n, t, d), dtype=torch.float32, device=x.device)
    _, _, _, ldy, _ = _ntd("out", out, dense_td=True)
    check(_lib.load().sagnn_layernorm_td_f32(x.data_ptr(), ld, ldt, n, t, d, _vec("gamma", gamma, d),
                                             _vec("beta", beta, d), float(eps), out.data_ptr(), ldy,
                                             _stream()))
    return out


def mhsa_mean(x: torch.Tensor, Wq, bq, Wk, bk, Wv, bv, heads: int, out: torch.Tensor | None = None):
    """MultiHeadSelfAttention + mean over T (reference Utils/attention.py:55-78, model.py:154-155):
    sagnn_mhsa_mean_f32. x [n, t, d] -> [n, d]."""
    n, t, d, ld, ldt = _ntd("x", x)
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=x.device)
    ldo = _f32_rows("out", out, d, n)
    lib = _lib.load()
    if _wide(d):
        ws = torch.empty(int(lib.sagnn_mhsa_wide_workspace_bytes(n, t, d)) // 4, dtype=torch.float32, device=x.device)
        check(lib.sagnn_mhsa_mean_wide_f32(
            x.data_ptr(), ld, ldt, n, t, d, int(heads), _vec("Wq", Wq, d * d), _vec("bq", bq, d),
            _vec("Wk", Wk, d * d), _vec("bk", bk, d), _vec("Wv", Wv, d * d), _vec("bv", bv, d),
            out.data_ptr(), ldo, ws.data_ptr(), ws.numel() * 4, _stream()))
        return out
    check(lib.sagnn_mhsa_mean_f32(
        x.data_ptr(), ld, ldt, n, t, d, int(heads), _vec("Wq", Wq, d * d), _vec("bq", bq, d),
        _vec("Wk", Wk, d * d), _vec("bk", bk, d), _vec("Wv", Wv, d * d), _vec("bv", bv, d),
        out.data_ptr(), ldo, _stream()))
    return out


def ln_mhsa_mean(x: torch.Tensor, gamma, beta, Wq, bq, Wk, bk, Wv, bv, heads: int, eps: float = 1e-12):
    """layer_norm over (T, d) -> MHSA -> mean (reference model.py:152-155) in one call:
    sagnn_ln_mhsa_mean_f32 (fused on the matrix-core path). x [n, t, d] -> [n, d]."""
    n, t, d, ld, ldt = _ntd("x", x)
    out = torch.empty((n, d), dtype=torch.float32, device=x.device)
    lib = _lib.load()
    need = int(lib.sagnn_ln_mhsa_mean_workspace_bytes(n, t, d, int(heads)))
    ws = torch.empty(need // 4, dtype=torch.float32, device=x.device) if need else None
    check(lib.sagnn_ln_mhsa_mean_f32(
        x.data_ptr(), ld, ldt, n, t, d, int(heads), _vec("gamma", gamma, d), _vec("beta", beta, d), float(eps),
        _vec("Wq", Wq, d * d), _vec("bq", bq, d), _vec("Wk", Wk, d * d), _vec("bk", bk, d), _vec("Wv", Wv, d * d),
        _vec("bv", bv, d), out.data_ptr(), d, _ptr(ws), need, _stream()))
    return out


def interval_fusion(x: torch.Tensor, p: dict, heads: int, out: torch.Tensor | None = None,
                    workspace: torch.Tensor | None = None):
    """LSTM -> layer_norm -> MHSA -> mean (reference model.py:135-155): sagnn_interval_fusion_f32.
    p: lstm_W [2d,4d], lstm_b [4d], ln_gamma [d], ln_beta [d], Wq/bq/Wk/bk/Wv/bv."""
    lib = _lib.load()
    n, t, d, ld, ldt = _ntd("x", x)
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=x.device)
    ldo = _f32_rows("out", out, d, n)
    need = int(lib.sagnn_interval_fusion_workspace_bytes(n, t, d))
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(max(need // 4, 1), dtype=torch.float32, device=x.device)
    check(lib.sagnn_interval_fusion_f32(
        x.data_ptr(), ld, ldt, n, t, d, int(heads), _vec("lstm_W", p["lstm_W"], 8 * d * d),
        _vec("lstm_b", p["lstm_b"], 4 * d), 1.0, _vec("ln_gamma", p["ln_gamma"], d),
        _vec("ln_beta", p["ln_beta"], d), 1e-12, _vec("Wq", p["Wq"], d * d), _vec("bq", p["bq"], d),
        _vec("Wk", p["Wk"], d * d), _vec("bk", p["bk"], d), _vec("Wv", p["Wv"], d * d),
        _vec("bv", p["bv"], d), out.data_ptr(), ldo, workspace.data_ptr(),
        workspace.numel() * workspace.element_size(), _stream()))
    return out


def dense_nn(x: torch.Tensor, W: torch.Tensor, bias: torch.Tensor | None = None, out: torch.Tensor | None = None,
             accumulate: bool = False):
    """Y (+)= X @ W + bias on the matrix cores (sagnn_dense_nn_f32). x [n, din] (row stride free),
    W [din, dout] contiguous."""
    n, din = int(x.shape[0]), int(x.shape[1])
    dout = int(W.shape[1])
    ldx = _f32_rows("x", x, din)
    if out is None:
        out = torch.empty((n, dout), dtype=torch.float32, device=x.device)
    ldy = _f32_rows("out", out, dout, n)
    check(_lib.load().sagnn_dense_nn_f32(x.data_ptr(), ldx, n, din, dout, _vec("W", W, din * dout),
                                         None if bias is None else _vec("bias", bias, dout), out.data_ptr(), ldy,
                                         int(accumulate), _stream()))
    return out


def dense_tn(x: torch.Tensor, g: torch.Tensor, dW: torch.Tensor, db: torch.Tensor | None = None):
    """dW += X^T @ G, db += colsum(G) (sagnn_dense_tn_f32). Accumulates: zero dW/db first."""
    n, din = int(x.shape[0]), int(x.shape[1])
    dout = int(g.shape[1])
    ldx = _f32_rows("x", x, din)
    ldg = _f32_rows("g", g, dout, n)
    check(_lib.load().sagnn_dense_tn_f32(x.data_ptr(), ldx, g.data_ptr(), ldg, n, din, dout,
                                         _vec("dW", dW, din * dout), None if db is None else _vec("db", db, dout),
                                         _stream()))
    return dW


def dense_tn_seg(x: torch.Tensor, g: torch.Tensor, dW: torch.Tensor, db: torch.Tensor | None = None):
    """dW += sum_s x[s]^T @ g[s], db += column sums of g (sagnn_dense_tn_seg_f32): x [s, n, din], g [s, n, dout] as
    VIEWS with any segment / row strides (unit column stride) — e.g. x.permute(1, 0, 2) of a node-major [n, t, d]
    against gate gradients stored [t, n, 4d]: a whole BPTT's weight gradient in one launch. Accumulates."""
    if x.dim() != 3 or g.dim() != 3 or x.shape[:2] != g.shape[:2]:
        raise ValueError(f"dense_tn_seg: need x [s, n, din] and g [s, n, dout], got {tuple(x.shape)} / {tuple(g.shape)}")
    for name, v in (("x", x), ("g", g)):
        if v.dtype != torch.float32 or not v.is_cuda or v.stride(2) != 1:
            raise ValueError(f"dense_tn_seg: {name} must be float32 on the GPU with unit column stride")
    s_, n, din = (int(v) for v in x.shape)
    dout = int(g.shape[2])
    if s_ == 0 or n == 0:
        return dW
    check(_lib.load().sagnn_dense_tn_seg_f32(x.data_ptr(), int(x.stride(1)), int(x.stride(0)), g.data_ptr(), int(g.stride(1)),
                                             int(g.stride(0)), n, s_, din, dout, _vec("dW", dW, din * dout),
                                             None if db is None else _vec("db", db, dout), _stream()))
    return dW


def mul(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor | None = None):
    """out = a * b element-wise (sagnn_mul_f32); contiguous float32 tensors of equal size."""
    if out is None:
        out = torch.empty_like(a)
    for name, x in (("a", a), ("b", b), ("out", out)):
        if x.dtype != torch.float32 or not x.is_contiguous() or x.numel() != a.numel():
            raise ValueError(f"{name}: need contiguous float32 tensors of equal size")
    check(_lib.load().sagnn_mul_f32(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()))
    return out


class Adam:
    """tf.train.AdamOptimizer with the reference's staircase exponential decay and L2 weights
    (model.py:245-250): state per parameter tensor, ONE sagnn_adam_multi_f32 launch per step.

    A parameter whose gradient is None still takes the step when it is L2-regularised: TF
    differentiates loss + reg*Regularize(), so timeEmbed and the dead [d, d] weights of
    model.py:81 receive 2*reg*w and decay. Un-regularised tensors without a gradient are left
    alone (TF's minimize skips variables with no gradient)."""

    def __init__(self, params: dict, lr: float, decay: float = 1.0, decay_step: int = 1, reg: float = 0.0,
                 reg_names=(), beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
        self.params = params
        self.lr0, self.decay, self.decay_step, self.reg = lr, decay, max(int(decay_step), 1), reg
        self.reg_names = set(reg_names)
        self.b1, self.b2, self.eps = beta1, beta2, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.global_step = 0

    def learning_rate(self) -> float:
        return self.lr0 * self.decay ** (self.global_step // self.decay_step)     # staircase=True

    def step(self, grads: dict):
        lr = self.learning_rate()
        self.global_step += 1
        names, keep = [], []
        for k, g in grads.items():
            if g is None and not (k in self.reg_names and self.reg != 0.0):
                continue
            p = self.params[k]
            if not p.is_contiguous():
                raise ValueError(f"parameter {k!r} must be contiguous")
            if g is not None:
                g = g.contiguous()
                if g.numel() != p.numel() or g.dtype != torch.float32:
                    raise ValueError(f"gradient of {k!r}: expected {p.numel()} float32 elements")
                keep.append(g)                      # alive until the launch is queued
            names.append((k, g))
        n = len(names)
        if n == 0:
            return
        P, G, M, V = ((ctypes.c_void_p * n)() for _ in range(4))
        C, L2 = (ctypes.c_int64 * n)(), (ctypes.c_float * n)()
        for i, (k, g) in enumerate(names):
            p = self.params[k]
            P[i], G[i], M[i], V[i] = p.data_ptr(), (None if g is None else g.data_ptr()), self.m[k].data_ptr(), self.v[k].data_ptr()
            C[i], L2[i] = p.numel(), (self.reg if k in self.reg_names else 0.0)
        check(_lib.load().sagnn_adam_multi_f32(n, P, G, M, V, C, L2, lr, self.b1, self.b2, self.eps, self.global_step,
                                               _stream()))

    def state_dict(self) -> dict:
        """Slots and step counter, as tf.train.Saver stores them with the variables (model.py:512-520)."""
        out = {"global_step": torch.tensor(self.global_step, dtype=torch.int64)}
        for k in self.params:
            out["m/" + k] = self.m[k].detach().cpu()
            out["v/" + k] = self.v[k].detach().cpu()
        return out

    def load_state_dict(self, state: dict):
        want = {"global_step"} | {"m/" + k for k in self.params} | {"v/" + k for k in self.params}
        if set(state) != want:
            raise KeyError(f"optimizer state keys differ: missing {sorted(want - set(state))[:4]}, "
                           f"unexpected {sorted(set(state) - want)[:4]}")
        for k in self.params:
            for slot, name in ((self.m, "m/"), (self.v, "v/")):
                if tuple(state[name + k].shape) != tuple(slot[k].shape):
                    raise ValueError(f"optimizer slot {name + k}: shape {tuple(state[name + k].shape)} != {tuple(slot[k].shape)}")
                slot[k].copy_(state[name + k])
        self.global_step = int(state["global_step"])


def leaky_add(a: torch.Tensor, b: torch.Tensor | None, leaky: float, out: torch.Tensor | None = None):
    """out = max(leaky*a, a) + b (sagnn_leaky_add_f32); contiguous float32."""
    if out is None:
        out = torch.empty_like(a)
    for name, x in (("a", a), ("b", b), ("out", out)):
        if x is not None and (x.dtype != torch.float32 or not x.is_contiguous() or x.numel() != a.numel()):
            raise ValueError(f"{name}: need contiguous float32 tensors of equal size")
    check(_lib.load().sagnn_leaky_add_f32(a.data_ptr(), _ptr(b), out.data_ptr(), float(leaky), a.numel(), _stream()))
    return out


def pair_score(U: torch.Tensor, I: torch.Tensor, uids: torch.Tensor, iids: torch.Tensor, S: torch.Tensor | None = None,
               A: torch.Tensor | None = None, locs: torch.Tensor | None = None, leaky: float = 1.0):
    """preds[e] = <U[uids[e]], I[iids[e]]> + <leaky(S[locs[e]]), A[iids[e]]> (sagnn_pair_score_f32)."""
    d = int(U.shape[1])
    n = int(uids.numel())
    out = torch.empty(n, dtype=torch.float32, device=U.device)
    for name, x in (("uids", uids), ("iids", iids), ("locs", locs)):
        if x is not None and (x.dtype != torch.int32 or not x.is_contiguous() or x.numel() != n):
            raise ValueError(f"{name}: need a contiguous int32 tensor of {n} elements")
    check(_lib.load().sagnn_pair_score_f32(
        U.data_ptr(), _f32_rows("U", U, d), I.data_ptr(), _f32_rows("I", I, d), _ptr(S),
        0 if S is None else _f32_rows("S", S, d), _ptr(A), 0 if A is None else _f32_rows("A", A, d),
        uids.data_ptr(), iids.data_ptr(), _ptr(locs), float(leaky), out.data_ptr(), n, d, _stream()))
    return out
