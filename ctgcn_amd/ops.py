"""Torch-facing wrappers over the C ABI (include/ctgcn_hip.h).  PyTorch is plumbing here: it owns
device memory, streams and the autograd graph; every kernel is in libctgcn_hip.so.

All entry points require CUDA (ROCm) tensors and raise otherwise — there is no CPU path.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, ptr


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.CtgcnHipError(
                "ctgcn_amd runs on MI355X only: got a %s tensor. Move inputs (features and CoreAdj) to the GPU; "
                "there is no CPU fallback." % t.device)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_launch_timer = None


def set_launch_timer(callback):
    """callback(name, start_event, end_event, meta) is invoked for every aggregation launch with HIP events
    recorded on the launch stream right before / after the kernel (bench.py's roofline measurement).
    None disables it (default)."""
    global _launch_timer
    _launch_timer = callback


class _timed(object):
    def __init__(self, name, **meta):
        self.name, self.meta = name, meta

    def __enter__(self):
        if _launch_timer is not None:
            self.start = torch.cuda.Event(enable_timing=True)
            self.end = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if _launch_timer is not None and exc[0] is None:
            self.end.record()
            _launch_timer(self.name, self.start, self.end, self.meta)
        return False


def _i32(t):
    return t if t.dtype == torch.int32 and t.is_contiguous() else t.to(torch.int32).contiguous()


# ------------------------------------------------------------------------------------------- ingest
def edges_to_csr(src, dst, w, n):
    """Edge rows (file order; CUDA int32/int32/float32 or None) -> (row_ptr, col, val) of the symmetric,
    de-duplicated (last row wins), zero-diagonal CSR — reference utils.py:23-58 graph semantics, on the GPU."""
    _need_cuda(src, dst, w)
    lib = _lib.load()
    src, dst = _i32(src), _i32(dst)
    if w is not None:
        w = w if (w.dtype == torch.float32 and w.is_contiguous()) else w.to(torch.float32).contiguous()
    m = src.numel()
    dev = src.device
    row_ptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    col = torch.empty(2 * m, dtype=torch.int32, device=dev)
    val = torch.empty(2 * m, dtype=torch.float32, device=dev)
    nnz = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        nbytes = lib.ctgcn_workspace_bytes(_lib.OP_INGEST, n, m, 0, 0)
        ws = torch.empty(max(nbytes, 1), dtype=torch.uint8, device=dev)
        check(lib.ctgcn_edges_to_csr(n, m, ptr(src), ptr(dst), ptr(w), ptr(row_ptr), ptr(col), ptr(val), ctypes.byref(nnz),
                                     ptr(ws), nbytes, _stream()), "ctgcn_edges_to_csr")
    k = int(nnz.value)
    return row_ptr, col[:k], val[:k]


# ------------------------------------------------------------------------------------------- k-core
def kcore(row_ptr, col, level_cap=-1):
    """Core number of every vertex (reference: networkx.core_number at structure_generation.py:35).
    row_ptr/col: symmetric CSR structure on the GPU.  Returns (core int32[n] on the GPU, max core).
    level_cap = L > 0 peels only levels below L and reports every core number >= L as L."""
    _need_cuda(row_ptr, col)
    lib = _lib.load()
    row_ptr, col = _i32(row_ptr), _i32(col)
    n = row_ptr.numel() - 1
    core = torch.empty(n, dtype=torch.int32, device=row_ptr.device)
    if n == 0:
        return core, 0
    with torch.cuda.device(row_ptr.device):
        nbytes = lib.ctgcn_workspace_bytes(_lib.OP_KCORE, n, col.numel(), 0, 0)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=row_ptr.device)
        mx = ctypes.c_int32(0)
        check(lib.ctgcn_kcore_i32(n, ptr(row_ptr), ptr(col), ptr(core), ptr(ws), nbytes, int(level_cap), ctypes.byref(mx),
                                  _stream()), "ctgcn_kcore_i32")
    return core, int(mx.value)


def edge_levels(row_ptr, col, val, core, hist_len):
    """level[e] = min(core[row], core[col]) plus the per-level entry count / weight sum."""
    _need_cuda(row_ptr, col, val, core)
    lib = _lib.load()
    n = row_ptr.numel() - 1
    dev = row_ptr.device
    level = torch.empty(col.numel(), dtype=torch.int32, device=dev)
    count = torch.zeros(hist_len, dtype=torch.int64, device=dev)
    wsum = torch.zeros(hist_len, dtype=torch.float64, device=dev)
    with torch.cuda.device(dev):
        check(lib.ctgcn_edge_levels_i32(n, ptr(row_ptr), ptr(col), ptr(val), ptr(core), ptr(level), ptr(count), ptr(wsum),
                                        hist_len, _stream()), "ctgcn_edge_levels_i32")
    return level, count, wsum


def slot_reorder(row_ptr, col, val, level, table, K):
    _need_cuda(row_ptr, col, val, level, table)
    lib = _lib.load()
    n = row_ptr.numel() - 1
    col2, val2 = torch.empty_like(col), torch.empty_like(val)
    slot2 = torch.empty(col.numel(), dtype=torch.uint8, device=col.device)
    with torch.cuda.device(col.device):
        check(lib.ctgcn_slot_reorder(n, K, ptr(row_ptr), ptr(col), ptr(val), ptr(level), ptr(table), table.numel(),
                                     ptr(col2), ptr(val2), ptr(slot2), _stream()), "ctgcn_slot_reorder")
    return col2, val2, slot2


# ------------------------------------------------------------------------------------- plain SpMM
def spmm_csr(row_ptr, col, val, x, out=None, accumulate=False):
    """Y = A·X (or Y += A·X): one torch.sparse.mm of layers.py:43/45."""
    _need_cuda(row_ptr, col, val, x)
    lib = _lib.load()
    if x.dtype != torch.float32:
        raise TypeError("fp32 features expected")
    x = x if x.stride(-1) == 1 else x.contiguous()
    n = row_ptr.numel() - 1
    d = x.shape[1]
    if out is None:
        out = torch.empty(n, d, dtype=torch.float32, device=x.device)
        accumulate = False
    with torch.cuda.device(x.device):
        check(lib.ctgcn_spmm_csr_f32(n, d, ptr(row_ptr), ptr(col), ptr(val), ptr(x), x.stride(0), ptr(out), out.stride(0),
                                     1 if accumulate else 0, _stream()), "ctgcn_spmm_csr_f32")
    return out


def _transpose_bias(src, bias):
    """out[n, d] = src[d, n]^T + bias[d] (bias may be None), row-major, one pass (ctgcn_transpose_bias_f32)"""
    lib = _lib.load()
    d, n = src.shape
    w = src if src.stride(1) == 1 else src.contiguous()
    out = torch.empty(n, d, dtype=torch.float32, device=src.device)
    with torch.cuda.device(src.device):
        check(lib.ctgcn_transpose_bias_f32(n, d, ptr(w), w.stride(0), ptr(bias.contiguous() if bias is not None else None),
                                           ptr(out), out.stride(0), _stream()), "ctgcn_transpose_bias_f32")
    return out


class _LinearOfIdentity(torch.autograd.Function):
    """weight^T + bias with autograd: d weight = (d out)^T is the same transpose the other way (the framework accumulates it through a
    strided copy at 0.4 TB/s — 2.4 ms per 1 M x 128 snapshot, and its forward leaves a column-major tensor the aggregation has to copy)."""

    @staticmethod
    def forward(ctx, weight, bias):
        ctx.has_bias = bias is not None
        return _transpose_bias(weight.detach(), None if bias is None else bias.detach())

    @staticmethod
    def backward(ctx, dout):
        if dout.shape[0] > 64 * 65535:                       # beyond the transpose kernel's grid: the framework's strided copy
            dw = dout.t().contiguous()
        else:
            dw = _transpose_bias(dout if dout.stride(1) == 1 else dout.contiguous(), None)  # [d, n] = dout[n, d]^T
        return dw, (dout.sum(0) if ctx.has_bias else None)


def linear_of_identity(weight, bias):
    """nn.Linear(weight, bias) applied to the N x N identity (one-hot node features): [N, out] = weight^T + bias, written
    row-major in one pass (weight.t() + bias in torch yields a column-major tensor that the next layer has to copy); differentiable."""
    _need_cuda(weight)
    if torch.is_grad_enabled() and (weight.requires_grad or (bias is not None and bias.requires_grad)):
        return _LinearOfIdentity.apply(weight, bias)
    return _transpose_bias(weight.detach(), None if bias is None else bias.detach())


# -------------------------------------------------------------------- CoreDiffusion aggregation
def _hub_pieces(lib, adj, long_rows, slots, d, device, transposed=False):
    """(hub_split, workspace, bytes) for hub rows long enough to be cut into pieces (several blocks per row + a fixed-order second pass)"""
    split = adj.hub_split(transposed) if long_rows is not None else 1
    if split <= 1:
        return 1, None, 0
    nbytes = int(lib.ctgcn_hub_workspace_bytes(long_rows.numel(), split, slots, d))
    return split, torch.empty(nbytes, dtype=torch.uint8, device=device), nbytes


def _aggregate_fwd(adj, x, relu):
    lib = _lib.load()
    n, d = x.shape
    H = torch.empty(n, adj.K, d, dtype=torch.float32, device=x.device)
    flags = adj.flags | (_lib.F_RELU if relu else 0)
    with torch.cuda.device(x.device), _timed("agg_fwd", n=n, d=d, K=adj.K, nnz=adj.nnz):
        long_rows = adj.long_rows()
        split, hub_ws, hub_bytes = _hub_pieces(lib, adj, long_rows, adj.K, d, x.device)
        check(lib.ctgcn_core_aggregate_f32(n, d, adj.K, ptr(adj.row_ptr), ptr(adj.col), ptr(adj.val), ptr(adj.slot), ptr(x),
                                           x.stride(0), ptr(H), flags, ptr(long_rows), 0 if long_rows is None else long_rows.numel(),
                                           adj.LONG_ROW, split, ptr(hub_ws), hub_bytes, _stream()), "ctgcn_core_aggregate_f32")
    return H


def _aggregate_gather(adj, Z, S0, relu):
    """dX[r] = S0[r] + sum_e val_e Z[col_e, slot_e] (ctgcn_core_aggregate_bwd_f32) for Z [n, K, d] / S0 [n, d] in matrix-row order"""
    lib = _lib.load()
    n, K, d = Z.shape
    flags = adj.flags | (_lib.F_RELU if relu else 0)
    dX = torch.empty(n, d, dtype=torch.float32, device=Z.device)
    t_ptr, t_col, t_val, t_slot = adj.transposed()
    long_rows = adj.long_rows(transposed=True)
    split, hub_ws, hub_bytes = _hub_pieces(lib, adj, long_rows, 1, d, Z.device, transposed=True)
    with torch.cuda.device(Z.device), _timed("agg_bwd", n=n, d=d, K=K, nnz=adj.nnz):
        check(lib.ctgcn_core_aggregate_bwd_f32(n, d, K, ptr(t_ptr), ptr(t_col), ptr(t_val), ptr(t_slot), ptr(Z), ptr(S0),
                                               ptr(dX), d, flags, ptr(long_rows), 0 if long_rows is None else long_rows.numel(),
                                               adj.LONG_ROW, split, ptr(hub_ws), hub_bytes, _stream()), "ctgcn_core_aggregate_bwd_f32")
    return dX


def _aggregate_bwd(adj, H, dH, relu):
    lib = _lib.load()
    n, K, d = dH.shape
    flags = adj.flags | (_lib.F_RELU if relu else 0)
    Z = torch.empty_like(dH)
    S0 = torch.empty(n, d, dtype=torch.float32, device=dH.device) if adj.self_loop else None
    with torch.cuda.device(dH.device), _timed("agg_bwd_prep", n=n, d=d, K=K, self_loop=adj.self_loop):
        check(lib.ctgcn_core_aggregate_bwd_prep_f32(n, d, K, ptr(dH), ptr(H), ptr(Z), ptr(S0), flags, _stream()),
              "ctgcn_core_aggregate_bwd_prep_f32")
    return _aggregate_gather(adj, Z, S0, relu)


class _CoreAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, relu):
        H = _aggregate_fwd(adj, x, relu)
        ctx.adj, ctx.relu = adj, relu
        ctx.save_for_backward(H)
        return H

    @staticmethod
    def backward(ctx, dH):
        (H,) = ctx.saved_tensors
        dH = dH.contiguous()
        return _aggregate_bwd(ctx.adj, H, dH, ctx.relu), None, None


def core_aggregate(x, adj, relu=True):
    """H[N, K, d] with H[:, j, :] = relu(sum_{i<=j} A_i x)  — layers.py:41-48 and the layout of :58."""
    _need_cuda(x, adj.col)
    if adj.K < 1:
        raise ValueError("empty k-core adjacency list")
    if x.dim() != 2 or x.shape[0] != adj.n:
        raise ValueError("features must be [%d, d], got %s" % (adj.n, tuple(x.shape)))
    if x.dtype != torch.float32:
        raise TypeError("fp32 features expected")
    if x.device != adj.device:
        raise ValueError("features on %s but adjacency on %s" % (x.device, adj.device))
    x = x if x.stride(1) == 1 else x.contiguous()
    return _CoreAggregate.apply(x, adj, relu)


# ------------------------------------------------------------- GRU over the core / time axis (+ sum, LayerNorm)
import os as _os
_GI_MAX_ELEMS = int(_os.environ.get("CTGCN_GI_MAX_ELEMS", 1 << 30))   # row-chunk bound: rows * steps * 4 * hidden elements (4 GiB of saved gates)


def gru_fused_ok(rnn, seq):
    """The HIP recurrent kernels cover the configuration every shipped CTGCN config uses: single-layer
    unidirectional batch_first GRU with hidden 128 on fp32 CUDA tensors (forward and backward)."""
    return (isinstance(rnn, torch.nn.GRU) and rnn.hidden_size == 128 and rnn.num_layers == 1 and not rnn.bidirectional
            and rnn.batch_first and seq.is_cuda and seq.dtype == torch.float32)


def _gru_bias(rnn, hid):
    """(b_ih + b_hh on r, z | b_ih on n,  b_hn): the two bias vectors the GRU kernels take.  Folded once per version of the two parameters
    (round 6: an AS-like forward spent 17 tiny torch kernels per window on re-folding them — clone, add, contiguous per GRU — and the
    fresh tensors changed the grouped launches' descriptors on every call); the folded vectors live in the operand-plane cache."""
    if not rnn.bias:
        return None, None
    b_ih, b_hh = rnn.bias_ih_l0, rnn.bias_hh_l0

    def fold():
        both = torch.empty(4 * hid, dtype=b_ih.dtype, device=b_ih.device)
        both[: 3 * hid] = b_ih.detach()
        both[: 2 * hid] += b_hh.detach()[: 2 * hid]           # r and z gates: both biases are simply added
        both[3 * hid:] = b_hh.detach()[2 * hid:]              # n gate: b_hn stays inside r * (W_hn h + b_hn)
        return both
    if plane_cache_enabled() and not b_ih.is_inference() and not b_hh.is_inference():
        tag = (b_hh.data_ptr(), b_hh._version)                # the cache validates b_ih (identity + version); b_hh rides along as a tag

        def make():
            both = fold()
            both._ctgcn_tag = tag
            return both, 16 * hid
        both = _plane_cache.get(b_ih, "gru_bias", make)
        if getattr(both, "_ctgcn_tag", None) != tag:
            _plane_cache.forget(b_ih, "gru_bias")
            both = _plane_cache.get(b_ih, "gru_bias", make)
    else:
        both = fold()
    return both[: 3 * hid], both[3 * hid:]


def _row_chunks(lib, rows, steps, hid):
    """Equal row chunks, multiples of the kernel's row granule (rows per block x CUs), bounded projection buffer."""
    if rows <= 0:
        return [(0, 0)]
    granule = int(lib.ctgcn_gru_row_granule())
    max_rows = max(granule, (_GI_MAX_ELEMS // (steps * 4 * hid)) // granule * granule)
    n_chunks = -(-rows // max_rows)
    chunk = -(-(-(-rows // n_chunks)) // granule) * granule
    return [(lo, min(chunk, rows - lo)) for lo in range(0, rows, chunk)]


def split_mfma_enabled():
    """fp32-accurate split arithmetic on the 16-bit matrix cores (default).  CTGCN_FP32_MFMA_ONLY=1 forces the
    plain fp32 paths (hipBLASLt fp32 GEMM + v_mfma_f32_16x16x4_f32 recurrence) for A/B comparisons."""
    import os
    return os.environ.get("CTGCN_FP32_MFMA_ONLY", "0") != "1"


def layer_kernel_enabled(reduce_sum=True):
    """ctgcn_gru_layer_f32: input projection and recurrence of a GRU with d_in = hidden = 128 in one kernel, both weight matrices
    resident on the CU (registers + LDS), the projection consumed from the MFMA accumulators (never materialised).  Bit-identical to
    the projection + recurrence kernel pair (tests/test_gpu_gru.py) at 1/7 of its HBM traffic.
    CTGCN_GRU_LAYER = 1 (default): both forms — the sum-over-steps form of CoreDiffusion (4.2-4.6 vs 6.4 ms per 1M x 8 call) and the
    per-step form of the temporal GRU (LayerNorm(h_t) emitted per unit through an LDS staging buffer: 11.8-12.4 vs 15.7 ms per 1M x 16 call).
    sum: the sum form only.  0: never.  CTGCN_GRU_LAYER_WAVES=4 selects the 4-wave builds (one wave per SIMD: 5.2 / 16.5 ms)."""
    import os
    mode = os.environ.get("CTGCN_GRU_LAYER", "1")
    if mode == "0":
        return False
    return bool(reduce_sum) or mode != "sum"


def forward_split_mode():
    """Arithmetic of the forward GRU products (include/ctgcn_hip.h CTGCN_SPLIT_*): 2 = fp16x2 (default: per-row scaled
    two-term fp16 split, three products — half the matrix work of bf16x3 and measured more accurate), 1 = bf16x3
    (CTGCN_GRU_SPLIT=bf16x3), 0 = fp32 MFMA (CTGCN_FP32_MFMA_ONLY=1)."""
    import os
    if not split_mfma_enabled():
        return 0
    return 1 if os.environ.get("CTGCN_GRU_SPLIT", "f16x2") == "bf16x3" else 2


_LINEAR_WS_MAX = 2 << 30       # bytes of fp16 planes per ctgcn_linear_f32 call (rows are chunked above that)


def linear_split_enabled():
    """CTGCN_LINEAR_SPLIT=0 (or CTGCN_FP32_MFMA_ONLY=1) selects the fp32 library GEMM instead of ctgcn_linear_f32."""
    import os
    return forward_split_mode() == 2 and os.environ.get("CTGCN_LINEAR_SPLIT", "1") != "0"


def linear_split_ok(x2d, weight):
    """ctgcn_linear_f32 covers fp32 CUDA operands with unit column stride and k >= 32 (any k and row stride: rows that are not
    16-byte aligned are read with scalar loads)."""
    return (linear_split_enabled() and x2d.is_cuda and x2d.dtype == torch.float32 and weight.dtype == torch.float32 and x2d.dim() == 2
            and x2d.stride(1) == 1 and weight.stride(1) == 1 and x2d.data_ptr() % 4 == 0 and weight.data_ptr() % 4 == 0
            and x2d.shape[0] > 0 and x2d.shape[1] >= 32 and x2d.stride(0) >= x2d.shape[1] and weight.stride(0) >= weight.shape[1])


class _PlaneCache(object):
    """Operand planes (ctgcn_split_rows_f32 / ctgcn_pack_weight_f32) of tensors that do not change between forwards: the weights of an
    inference run and node features the caller has marked static (mark_static: the reference builds them once and feeds them to every
    batch, train.py:72-76 -> embedding.py:318).  Keyed by storage address + shape + strides + kind, validated by identity (weak
    reference) and by the tensor's version counter: an in-place update — optimizer.step(), load_state_dict(), p.copy_() — re-splits.
    What the version counter does NOT see: writes through `.data` / raw pointers (p.data.mul_(2), dist.broadcast(p.data)).  After such
    an edit call ops.invalidate_plane_cache() (snapshot_parallel.shard_cgcn does).  Tensors created under torch.inference_mode() have no
    version counter: they are split on every call and never cached.  Entries die with their tensor; at most `limit` bytes are kept
    (least recently used first out)."""

    def __init__(self, limit=48 << 30):
        import collections
        self.limit, self.bytes, self.entries = limit, 0, collections.OrderedDict()
        self.capture_hold = None                 # a list while a frozen-weights capture runs (graph_capture.GraphedInference)

    def _drop(self, key):
        e = self.entries.pop(key, None)
        if e is not None:
            self.bytes -= e[3]

    def clear(self):
        self.entries.clear()
        self.bytes = 0

    def forget(self, t, kind):
        self._drop((t.data_ptr(), tuple(t.shape), tuple(t.stride()), str(t.device), kind))

    def get(self, t, kind, make):
        """make() -> (buffer, nbytes): the operand form `kind` of tensor t, built on the current stream"""
        import weakref
        capturing = torch.cuda.is_current_stream_capturing()
        if t.is_inference() or (capturing and self.capture_hold is None):
            # no version counter to validate against (t._version raises): never cached.  Under hipGraph capture the operand form is built
            # INSIDE the graph, from the tensor's storage at replay time: a replay sees in-place weight updates (graph_capture.py's contract)
            return make()[0]
        key = (t.data_ptr(), tuple(t.shape), tuple(t.stride()), str(t.device), kind)
        e = self.entries.get(key)
        cur = torch.cuda.current_stream(t.device)
        if e is not None and e[0]() is t and e[1] == t._version:
            self.entries.move_to_end(key)
            if capturing:
                # GraphedInference(frozen_weights=True): the graph reads the operand form built by the warm-up forwards (the device was
                # synchronised before the capture began).  The runner keeps the buffer alive for as long as the graph exists.
                self.capture_hold.append(e[2])
                return e[2]
            if e[5] != cur:                      # built on another stream: order behind it, keep the buffer alive for this stream too
                cur.wait_event(e[4])
                e[2].record_stream(cur)
            return e[2]
        if capturing:
            return make()[0]                     # a miss under capture: built in-graph, nothing kept (the buffer belongs to the capture's pool)
        self._drop(key)
        buf, nbytes = make()
        ev = torch.cuda.Event()
        ev.record(cur)
        self.entries[key] = (weakref.ref(t, lambda _r, key=key: self._drop(key)), t._version, buf, nbytes, ev, cur)
        self.bytes += nbytes
        while self.bytes > self.limit and len(self.entries) > 1:
            self._drop(next(iter(self.entries)))
        return buf

    def planes(self, t, lib):
        """row-major fp16 planes + row scales of t [rows, k] (the A operand form of the split GEMM)"""
        def make():
            rows, k = t.shape
            nbytes = int(lib.ctgcn_split_planes_bytes(rows, k))
            buf = torch.empty(nbytes, dtype=torch.uint8, device=t.device)
            check(lib.ctgcn_split_rows_f32(rows, k, ptr(t), t.stride(0), ptr(buf), nbytes, _stream()), "ctgcn_split_rows_f32")
            return buf, nbytes
        return self.get(t, "planes", make)

    def packed(self, t, lib):
        """packed W operand of the weight t [n_out, k] (ctgcn_pack_weight_f32: the split in matrix-core fragment order + column scales)"""
        def make():
            n_out, k = t.shape
            nbytes = int(lib.ctgcn_pack_weight_bytes(n_out, k))
            buf = torch.empty(nbytes, dtype=torch.uint8, device=t.device)
            check(lib.ctgcn_pack_weight_f32(n_out, k, ptr(t), t.stride(0), ptr(buf), nbytes, _stream()), "ctgcn_pack_weight_f32")
            return buf, nbytes
        return self.get(t, "packed", make)


_plane_cache = _PlaneCache()


def invalidate_plane_cache():
    """Forget every cached operand plane.  Needed after weights or static features were edited behind the version counter's back
    (writes through `.data`, raw pointers, collectives on `.data`)."""
    _plane_cache.clear()


def mark_static(x):
    """Declare that the feature tensor x is fed to every forward unchanged (what the reference's loader output is: built once in
    train.py:72-76, passed to every batch, embedding.py:318): the first dense Linear keeps x's operand planes between forwards instead of
    splitting x on every call.  ctgcn_amd.helper.DataLoader marks what it returns; in-place edits through the autograd-visible API
    re-split (version counter), edits through `.data` need invalidate_plane_cache().  Returns x."""
    x._ctgcn_static = True
    return x


def is_static(x):
    return bool(getattr(x, "_ctgcn_static", False))


def plane_cache_enabled():
    """CTGCN_PLANE_CACHE=0: ctgcn_linear_f32 splits both operands on every call (round 3's behaviour) for A/B runs."""
    import os
    return os.environ.get("CTGCN_PLANE_CACHE", "1") != "0"      # (under hipGraph capture _PlaneCache.get builds in-graph and keeps nothing)


class Planes(object):
    """the operand form of a dense activation [rows, k] that never existed as fp32 rows: per-row scale + two fp16 planes, written by the GEMM
    that produced it (ctgcn_linear_packed_chain_f32) for the GEMM that consumes it"""
    __slots__ = ("buf", "rows", "k", "device")

    def __init__(self, buf, rows, k):
        self.buf, self.rows, self.k, self.device = buf, rows, k, buf.device

    @property
    def shape(self):
        return (self.rows, self.k)


def _linear_chunk_rows(k):
    kp = -(-k // 64) * 64
    return max(128, (_LINEAR_WS_MAX // (kp * 4 + 4)) // 128 * 128)


def mlp_chain_enabled():
    """CTGCN_MLP_CHAIN=0: every Linear of an MLP writes fp32 rows and the next one splits them again (A/B runs)"""
    import os
    return os.environ.get("CTGCN_MLP_CHAIN", "1") != "0"


def linear_chain_ok(x, weight, next_weight):
    """Inference through two consecutive dense Linear layers (layers.py:95-106): the first one's output can leave as the second one's operand
    planes (linear_split(..., planes_out=True)) — both on the split GEMM in one row chunk, the hidden width at most 512."""
    if not (mlp_chain_enabled() and linear_split_enabled() and plane_cache_enabled()):
        return False
    if isinstance(x, Planes):
        rows, k = x.rows, x.k
        if not (weight.is_cuda and weight.dtype == torch.float32 and weight.stride(1) == 1 and weight.data_ptr() % 4 == 0 and weight.stride(0) >= k):
            return False
    else:
        if not (torch.is_tensor(x) and not x.is_sparse and x.dim() == 2 and linear_split_ok(x, weight)):
            return False
        rows, k = x.shape
    hidden = weight.shape[0]
    if weight.shape[1] != k or next_weight.shape[1] != hidden or hidden < 32 or hidden > 512:
        return False
    if not (next_weight.is_cuda and next_weight.dtype == torch.float32 and next_weight.dim() == 2 and next_weight.stride(1) == 1
            and next_weight.data_ptr() % 4 == 0 and next_weight.stride(0) >= next_weight.shape[1] and next_weight.device == weight.device
            and weight.device == x.device):
        return False
    return rows > 0 and rows <= _linear_chunk_rows(k) and rows <= _linear_chunk_rows(hidden)


def linear_split(x2d, weight, bias, out=None, selu=False, static_x=False, planes_out=False):
    """out[rows, n_out] = x2d @ weight^T + bias in fp32-accurate fp16x2 split arithmetic on the matrix cores (gemm_h2_panel_kernel);
    selu: F.selu applied in the GEMM's epilogue (one pass over the output less).
    The weight's packed operand is built once per weight version (_PlaneCache); static_x: x2d is a tensor the caller feeds to every
    forward unchanged (mark_static): its planes are kept too.
    x2d may be a Planes object (the output of a call with planes_out=True: the hidden activations of an MLP go from GEMM to GEMM as operand
    planes, never as fp32 rows; linear_chain_ok says when)."""
    lib = _lib.load()
    rows, k = x2d.shape
    n_out = weight.shape[0]
    chunk = _linear_chunk_rows(k)
    b = None if bias is None else bias.detach().contiguous()
    w = weight.detach()
    act = _lib.ACT_SELU if selu else _lib.ACT_NONE
    if isinstance(x2d, Planes) or planes_out:
        if not (rows <= chunk and plane_cache_enabled()) or (planes_out and n_out > 512):
            raise ValueError("linear_split: operand planes in / out need one row chunk, the plane cache and n_out <= 512 (linear_chain_ok)")
        dev = x2d.device
        with torch.cuda.device(dev):
            wp = _plane_cache.packed(weight, lib)
            if isinstance(x2d, Planes):
                xp = x2d.buf
            elif static_x and not x2d.requires_grad:
                xp = _plane_cache.planes(x2d, lib)
            else:
                nbytes = int(lib.ctgcn_split_planes_bytes(rows, k))
                xp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                with _timed("linear_aux", rows=rows, k=k, split=True):
                    check(lib.ctgcn_split_rows_f32(rows, k, ptr(x2d), x2d.stride(0), ptr(xp), nbytes, _stream()), "ctgcn_split_rows_f32")
            if planes_out:
                obytes = int(lib.ctgcn_split_planes_bytes(rows, n_out))
                obuf = torch.empty(obytes, dtype=torch.uint8, device=dev)
                with _timed("linear_split", rows=rows, k=k, n_out=n_out, planes=True, chain=True):
                    check(lib.ctgcn_linear_packed_chain_f32(rows, n_out, k, ptr(xp), ptr(wp), ptr(b), act, ptr(obuf), obytes, _stream()),
                          "ctgcn_linear_packed_chain_f32")
                return Planes(obuf, rows, n_out)
            if out is None:
                out = torch.empty(rows, n_out, dtype=torch.float32, device=dev)
            with _timed("linear_split", rows=rows, k=k, n_out=n_out, planes=True):
                check(lib.ctgcn_linear_packed_f32(rows, n_out, k, ptr(xp), ptr(wp), ptr(b), act, ptr(out), out.stride(0), _stream()), "ctgcn_linear_packed_f32")
        return out
    if out is None:
        out = torch.empty(rows, n_out, dtype=torch.float32, device=x2d.device)
    if rows <= chunk and plane_cache_enabled():
        with torch.cuda.device(x2d.device):
            wp = _plane_cache.packed(weight, lib)
            if static_x and not x2d.requires_grad:
                xp = _plane_cache.planes(x2d, lib)
            else:
                nbytes = int(lib.ctgcn_split_planes_bytes(rows, k))
                xp = torch.empty(nbytes, dtype=torch.uint8, device=x2d.device)
                with _timed("linear_aux", rows=rows, k=k, split=True):
                    check(lib.ctgcn_split_rows_f32(rows, k, ptr(x2d), x2d.stride(0), ptr(xp), nbytes, _stream()), "ctgcn_split_rows_f32")
            with _timed("linear_split", rows=rows, k=k, n_out=n_out, planes=True):
                check(lib.ctgcn_linear_packed_f32(rows, n_out, k, ptr(xp), ptr(wp), ptr(b), act, ptr(out), out.stride(0), _stream()), "ctgcn_linear_packed_f32")
        return out
    with torch.cuda.device(x2d.device):
        ws_bytes = int(lib.ctgcn_linear_workspace_bytes(min(rows, chunk), n_out, k))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x2d.device)
        for lo in range(0, rows, chunk):
            n = min(chunk, rows - lo)
            xs, ys = x2d[lo:lo + n], out[lo:lo + n]
            with _timed("linear_split", rows=n, k=k, n_out=n_out):
                check(lib.ctgcn_linear_f32(n, n_out, k, ptr(xs), xs.stride(0), ptr(w), w.stride(0), ptr(b), act,
                                           ptr(ys), ys.stride(0), ptr(ws), ws_bytes, _stream()), "ctgcn_linear_f32")
    return out


def linear_train_enabled():
    """CTGCN_LINEAR_TRAIN=0: nn.Linear under autograd stays torch's (fp32 library GEMMs) for A/B runs"""
    import os
    return os.environ.get("CTGCN_LINEAR_TRAIN", "1") != "0"


class _LinearSplit(torch.autograd.Function):
    """nn.Linear on dense rows under autograd (layers.py:95-106 in training: the MLP of CTGCN-S, 1 737 -> 500 -> 500 -> 128 on the
    Facebook-like config): y = x W^T + b and dx = dy W on the split GEMM (the forward's arithmetic), dW = dy^T x — a product that contracts
    over the ROWS, which the panel kernel does not do — and db stay with the library.  Facebook-like training step: the fp32 library GEMMs
    of the three layers were 88 of 139 ms."""

    @staticmethod
    def forward(ctx, x, weight, bias, static_x):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return linear_split(x, weight, bias, static_x=static_x)       # x itself: the plane cache knows a static x by identity

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:                                # dx = dy @ w_t^T with w_t = weight^T [k, n_out], transposed + packed once per version
            dx = linear_split(dy, _transposed(weight), None) if _transposed_split_ok(dy, weight) else dy @ weight.detach()
        if ctx.needs_input_grad[1]:
            dw = torch.mm(dy.t(), x.detach())
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return dx, dw, db, None


def linear_split_train(x2d, weight, bias, static_x=False):
    return _LinearSplit.apply(x2d, weight, bias, bool(static_x))


def _project(x2d, w_ih, bias, out, steps_blocked=0):
    """out = x2d @ w_ih^T + bias — the GRU input projection.  Returns True when `out` was written in the recurrence
    kernel's blocked tile layout (only asked for with steps_blocked > 0, only done by the fp16x2 kernel), else [rows, 3h]."""
    mode = forward_split_mode()
    if mode and x2d.shape[1] == 128 and w_ih.shape[0] == 384 and x2d.stride(1) == 1 \
            and x2d.stride(0) % 4 == 0 and x2d.data_ptr() % 16 == 0 and w_ih.is_contiguous():
        lib = _lib.load()
        blocked = steps_blocked if mode == 2 else 0
        with _timed("gru_proj", rows=x2d.shape[0]):
            check(lib.ctgcn_gru_input_proj_f32(x2d.shape[0], 128, 128, ptr(x2d), x2d.stride(0), ptr(w_ih), ptr(bias), ptr(out),
                                               mode, blocked, _stream()), "ctgcn_gru_input_proj_f32")
        return blocked > 0
    out = out[: x2d.shape[0] * w_ih.shape[0]].view(x2d.shape[0], w_ih.shape[0])
    if linear_split_ok(x2d, w_ih):
        linear_split(x2d, w_ih, bias, out=out)
        return False
    if bias is None:
        torch.mm(x2d, w_ih.t(), out=out)
    else:
        torch.addmm(bias, x2d, w_ih.t(), out=out)
    return False


def _gi_buffer(rows, steps, hid, device):
    """flat fp32 buffer for the projection of `rows` sequences; whole 64-row tiles (the blocked layout writes by tile)"""
    return torch.empty(-(-rows // 64) * 64 * steps * 3 * hid, dtype=torch.float32, device=device)


def _transposed(w):
    """w^T as a contiguous tensor, built once per version of w (every backward re-transposed the weight and pushed the temporary
    through the packed-operand cache, which keys on the tensor object — an entry added and dropped per call).  The cached transpose is a
    stable object, so its packed form (linear_split -> _PlaneCache.packed) is kept with it and dies with it when w changes."""
    def make():
        t = w.detach().t().contiguous()
        return t, t.numel() * t.element_size()
    if not plane_cache_enabled():
        return make()[0]
    return _plane_cache.get(w, "transposed", make)


def _transposed_split_ok(x2d, w):
    """linear_split_ok(x2d, w^T contiguous) without building the transpose"""
    return (linear_split_enabled() and x2d.is_cuda and w.is_cuda and x2d.device == w.device and x2d.dtype == torch.float32 and w.dtype == torch.float32
            and x2d.dim() == 2 and w.dim() == 2 and x2d.stride(1) == 1 and x2d.data_ptr() % 4 == 0 and x2d.shape[0] > 0 and x2d.shape[1] >= 32
            and x2d.stride(0) >= x2d.shape[1] and x2d.shape[1] == w.shape[0])


def _project_grad(dgi, w_ih, out):
    """out[rows, d_in] = dgi[rows, 3h] @ w_ih — gradient of the input projection w.r.t. its input."""
    if split_mfma_enabled() and w_ih.shape == (384, 128) and w_ih.is_contiguous() and dgi.is_contiguous() \
            and out.stride(1) == 1 and out.stride(0) % 4 == 0 and out.data_ptr() % 16 == 0:
        lib = _lib.load()
        check(lib.ctgcn_gru_input_grad_f32(dgi.shape[0], 128, 128, ptr(dgi), ptr(w_ih.detach()), ptr(out), out.stride(0), _stream()),
              "ctgcn_gru_input_grad_f32")
        return
    if _transposed_split_ok(dgi, w_ih) and out.stride(1) == 1:
        # the 500-wide first layer (layers.py:59 with input_size = hid_dim): the fp32 library GEMM took 1.8 ms per Enron-like snapshot,
        # 14 % of the training step; split of dgi (0.3 ms) + gemm_h2_panel_kernel (0.45 ms).  Same fp32-accurate arithmetic as the forward.
        # w_t [d_in, 3h] is the "weight" of out = dgi @ w_t^T: transposed and packed once per weight version (_transposed)
        linear_split(dgi, _transposed(w_ih), None, out=out)
        return
    torch.mm(dgi, w_ih.detach(), out=out)


def keep_projection_enabled():
    """CTGCN_KEEP_GI=0: the backward of a GRU with d_in != 128 always recomputes its input projection (A/B runs)"""
    import os
    return os.environ.get("CTGCN_KEEP_GI", "1") != "0"


def wide_weight_grad_enabled():
    """CTGCN_WIDE_DW=0: dW_ih of a GRU with d_in > 128 stays an fp32 library GEMM (A/B runs)"""
    import os
    return os.environ.get("CTGCN_WIDE_DW", "1") != "0"


_DW_PAIRS = 128          # block pairs of ctgcn_gru_weight_grad_f32: 256 blocks = one per CU of an MI355X


def _weight_grad(part, g01, g2, x2d, steps, shift, accumulate):
    """part[p] (+)= block pair p's share of  sum_r [g01[r, :2h] | g2[r, :h]]^T x'[r]  (x' = x2d, or x2d shifted one step
    inside each sequence); part.sum(0) is the weight gradient."""
    lib = _lib.load()
    check(lib.ctgcn_gru_weight_grad_f32(g01.shape[0], steps, 128, ptr(g01), g01.stride(0), ptr(g2), g2.stride(0), ptr(x2d),
                                        x2d.stride(0), 1 if shift else 0, ptr(part), part.shape[0], 1 if accumulate else 0,
                                        _stream()), "ctgcn_gru_weight_grad_f32")


def _gru_forward(seq, w_ih, w_hh, bias, b_hn, ln_w, ln_b, eps, reduce_sum, out=None, keep_gi=None):
    """keep_gi: a list that receives (gi buffer, blocked layout flag) when the projection was materialised in one chunk (training: the
    backward's recompute pass starts from it instead of splitting x and multiplying by W_ih again)"""
    lib = _lib.load()
    rows, steps, d_in = seq.shape
    hid = w_hh.shape[1]
    if out is None:
        out = torch.empty((rows, hid) if reduce_sum else (rows, steps, hid), dtype=torch.float32, device=seq.device)
    elif not (reduce_sum and out.shape == (rows, hid) and out.dtype == torch.float32 and out.stride(1) == 1
              and out.stride(0) % 2 == 0 and out.stride(0) >= hid and out.device == seq.device):
        raise ValueError("gru: out must be a [rows, %d] fp32 view with unit column stride (reduce_sum only)" % hid)
    ldo = out.stride(0) if reduce_sum else 0
    if rows == 0:
        return out
    split = forward_split_mode()
    if split == 2 and d_in == hid and layer_kernel_enabled(reduce_sum) and seq.stride(2) == 1 and seq.stride(1) % 4 == 0 \
            and seq.stride(0) == steps * seq.stride(1) and seq.data_ptr() % 16 == 0 and w_ih.is_contiguous():
        # projection + recurrence in one kernel, both weight matrices in the register file, gi never materialised
        with torch.cuda.device(seq.device), _timed("gru_layer", rows=rows, steps=steps, reduce_sum=bool(reduce_sum)):
            check(lib.ctgcn_gru_layer_f32(rows, steps, d_in, hid, ptr(seq), seq.stride(1), ptr(w_ih), ptr(w_hh), ptr(bias), ptr(b_hn),
                                          ptr(ln_w), ptr(ln_b), eps, 1 if reduce_sum else 0, ptr(out), ldo, None, None, 0, _stream()), "ctgcn_gru_layer_f32")
        return out
    chunks = _row_chunks(lib, rows, steps, hid)
    gi_buf = _gi_buffer(chunks[0][1], steps, hid, seq.device)
    with torch.cuda.device(seq.device):
        for lo, n in chunks:
            blocked = _project(seq[lo:lo + n].reshape(n * steps, d_in), w_ih, bias, gi_buf, steps_blocked=steps)
            with _timed("gru_seq", rows=n, steps=steps):
                check(lib.ctgcn_gru_seq_f32(n, steps, hid, ptr(gi_buf), ptr(w_hh), ptr(b_hn), ptr(ln_w), ptr(ln_b), eps,
                                            1 if reduce_sum else 0, ptr(out[lo:lo + n]), ldo, None, split, 1 if blocked else 0,
                                            None, None, None, _stream()), "ctgcn_gru_seq_f32")
        if keep_gi is not None and len(chunks) == 1:
            keep_gi.append((gi_buf, blocked))
    return out


def gru_steps_scattered_ok(rnn, base):
    """gru_sequence_scattered covers what the register-resident layer kernel covers: GRU 128 -> 128, fp16x2 arithmetic, 8-wave build"""
    import os
    return (gru_fused_ok(rnn, base) and rnn.input_size == rnn.hidden_size and forward_split_mode() == 2 and layer_kernel_enabled(False)
            and os.environ.get("CTGCN_GRU_LAYER_WAVES", "8") != "4" and rnn.weight_ih_l0.is_contiguous()
            and not (torch.is_grad_enabled() and any(p.requires_grad for p in rnn.parameters())))


def gru_sequence_scattered(rnn, norm, base, step_offsets, ld_row, rows):
    """LayerNorm(GRU(x)) -> [rows, steps, 128] (models.py:249-250) for an input whose steps live at base + step_offsets[t] + r * ld_row
    (floats) — e.g. the receive buffer of the snapshot-parallel exchange — without gathering them into a [rows, steps, 128] tensor first.
    Inference only; same kernel and arithmetic as gru_sequence(..., reduce_sum=False)."""
    lib = _lib.load()
    steps, hid = int(step_offsets.numel()), rnn.hidden_size
    bias, b_hn = _gru_bias(rnn, hid)
    ln_w = None if norm is None else norm.weight
    ln_b = None if norm is None else norm.bias
    eps = 0.0 if norm is None else float(norm.eps)
    out = torch.empty(rows, steps, hid, dtype=torch.float32, device=base.device)
    if rows == 0:
        return out
    with torch.cuda.device(base.device), _timed("gru_layer", rows=rows, steps=steps, reduce_sum=False):
        check(lib.ctgcn_gru_layer_f32(rows, steps, hid, hid, ptr(base), hid, ptr(rnn.weight_ih_l0.detach()), ptr(rnn.weight_hh_l0.detach().contiguous()),
                                      ptr(bias), ptr(b_hn), ptr(ln_w), ptr(ln_b), eps, 0, ptr(out), 0, None, ptr(step_offsets), int(ld_row), _stream()),
              "ctgcn_gru_layer_f32")
    return out


def aggregate_split_enabled():
    """CTGCN_AGG_SPLIT=0 keeps aggregation and GRU input projection apart (fp32 H in between) for A/B runs."""
    import os
    return linear_split_enabled() and os.environ.get("CTGCN_AGG_SPLIT", "1") != "0"


_AGG_SPLIT_MAX = 16 << 30     # bytes of fp16 planes the fused aggregation may write in one call (config 5: 4.1 GB per snapshot)


def aggregate_split_ok(rnn, x, adj):
    """Inference through a CoreDiffusion layer: the aggregation kernel can hand its consumer fp16 planes + row scales instead of
    fp32 rows (ctgcn_core_aggregate_split_f32) — the split GEMM of the GRU input projection when d_in != 128, the
    register-resident GRU layer kernel (ctgcn_gru_layer_presplit_f32) when d_in = hidden = 128."""
    if not aggregate_split_enabled() or x.dim() != 2 or not gru_fused_ok(rnn, x) or x.shape[0] != adj.n or x.device != adj.device:
        return False
    if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in rnn.parameters())):
        return False
    d, hid = x.shape[1], rnn.hidden_size
    if d % 4 or d < 32 or d > 512 or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16 or adj.K < 1 or adj.n < 1:
        return False
    if d == hid:        # the layer kernel path
        return layer_kernel_enabled(True) and adj.n * adj.K * (d * 4 + 4) <= _AGG_SPLIT_MAX and rnn.weight_ih_l0.is_contiguous()
    if adj.n * adj.K * (-(-d // 64) * 64 * 4 + 4) > _LINEAR_WS_MAX:
        return False
    return len(_row_chunks(_lib.load(), adj.n, adj.K, hid)) == 1


def row_plan_enabled():
    """CTGCN_DEDUP=0: the inference path writes and multiplies every (node, core) row of H, repeated or not (A/B runs)."""
    import os
    return os.environ.get("CTGCN_DEDUP", "1") != "0"


def aggregate_split_planes(x, adj, n_out, plan=None, ws=None):
    """ctgcn_core_aggregate_split_f32: relu(cumulative A_k x) as fp16 planes + row scales in a workspace (returned).  n_out = width of
    the GEMM that follows (1: the GRU layer kernel).  plan = adj.row_plan(tile): repeated rows are not written — tile 16 with the layer
    kernel as consumer (holes in the [n K] row layout), tile 64 with the GEMM (compact operand rows, plan["operand_rows"] of them)."""
    lib = _lib.load()
    n, d = x.shape
    flags = adj.flags | _lib.F_RELU
    long_rows = adj.long_rows()
    n_long = 0 if long_rows is None else long_rows.numel()
    ws_bytes = int(lib.ctgcn_core_aggregate_split_workspace_bytes(n, d, adj.K, n_out, n_long))
    if ws is None:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    compact = plan is not None and not (d == 128 and n_out == 1)
    if plan is not None and plan["tile"] != (adj.PLAN_TILE_GEMM if compact else adj.PLAN_TILE):
        raise ValueError("aggregate_split_planes: plan tile %d does not belong to this consumer" % plan["tile"])
    hub_dest = adj.plan_row_dest(plan, long_rows, compact) if (plan is not None and n_long) else None
    split, hub_ws, hub_bytes = _hub_pieces(lib, adj, long_rows, adj.K, d, x.device)
    with _timed("agg_fwd", n=n, d=d, K=adj.K, nnz=adj.nnz, split=True, rows_written=(plan["new_rows"] if plan is not None else n * adj.K)):
        check(lib.ctgcn_core_aggregate_split_f32(n, d, adj.K, ptr(adj.row_ptr), ptr(adj.col), ptr(adj.val), ptr(adj.slot), ptr(x), x.stride(0),
                                                 flags, ptr(long_rows), n_long, adj.LONG_ROW, n_out,
                                                 ptr(plan["order"]) if plan is not None else None, ptr(plan["tile_mask"]) if plan is not None else None,
                                                 ptr(plan["tile_base"]) if compact else None, plan["operand_rows"] if compact else 0,
                                                 ptr(hub_dest), split, ptr(hub_ws), hub_bytes, ptr(ws), ws_bytes, _stream()), "ctgcn_core_aggregate_split_f32")
    return ws, ws_bytes


def gru_layer_presplit(ws, n, K, rnn, norm, out, plan=None):
    """ctgcn_gru_layer_presplit_f32 on the planes aggregate_split_planes(x, adj, 1, plan) wrote: LayerNorm(sum_k GRU(H)_k) -> out."""
    lib = _lib.load()
    hid = rnn.hidden_size
    bias, b_hn = _gru_bias(rnn, hid)
    w_ih, w_hh = rnn.weight_ih_l0.detach(), rnn.weight_hh_l0.detach().contiguous()
    ln_w = None if norm is None else norm.weight
    ln_b = None if norm is None else norm.bias
    eps = 0.0 if norm is None else float(norm.eps)
    with _timed("gru_layer", rows=n, steps=K, reduce_sum=True, presplit=True, new_rows=(plan["new_rows"] if plan is not None else n * K)):
        check(lib.ctgcn_gru_layer_presplit_f32(n, K, hid, ptr(ws), ptr(w_ih), ptr(w_hh), ptr(bias), ptr(b_hn), ptr(ln_w), ptr(ln_b), eps,
                                               ptr(out), out.stride(0), ptr(plan["order"]) if plan is not None else None,
                                               ptr(plan["tile_mask"]) if plan is not None else None, _stream()), "ctgcn_gru_layer_presplit_f32")
    return out


def core_diffusion_split(x, adj, rnn, norm, out=None):
    """LayerNorm(sum_k GRU(relu(cumulative A_k x))_k) — CoreDiffusion.forward (layers.py:41-62) for inference, no fp32 H:
    aggregation -> fp16 planes + row scales, then  d_in = hidden = 128: the register-resident GRU layer kernel reads the planes
    (under the graph's row plan: rows of H that repeat the row before are neither written nor multiplied by W_ih again);
    d_in != 128: split GEMM -> gate pre-activations -> recurrence + sum + LayerNorm kernel."""
    lib = _lib.load()
    n, d = x.shape
    K, hid = adj.K, rnn.hidden_size
    n_out = 3 * hid
    if out is None:
        out = torch.empty(n, hid, dtype=torch.float32, device=x.device)
    elif not (out.shape == (n, hid) and out.dtype == torch.float32 and out.stride(1) == 1 and out.stride(0) % 2 == 0
              and out.stride(0) >= hid and out.device == x.device):
        raise ValueError("core_diffusion_split: out must be a [rows, %d] fp32 view with unit column stride" % hid)
    with torch.cuda.device(x.device):
        if d == hid:                                       # -> ctgcn_gru_layer_presplit_f32; its weights need no plane workspace
            plan = adj.row_plan() if row_plan_enabled() else None
            ws, _ = aggregate_split_planes(x, adj, 1, plan)
            return gru_layer_presplit(ws, n, K, rnn, norm, out, plan)
        bias, b_hn = _gru_bias(rnn, hid)
        w_ih, w_hh = rnn.weight_ih_l0.detach(), rnn.weight_hh_l0.detach().contiguous()
        ln_w = None if norm is None else norm.weight
        ln_b = None if norm is None else norm.bias
        eps = 0.0 if norm is None else float(norm.eps)
        plan = adj.row_plan(adj.PLAN_TILE_GEMM) if (row_plan_enabled() and forward_split_mode() == 2) else None
        rows = plan["operand_rows"] if plan is not None else n * K          # under a plan the GEMM only sees rows that bring a new x
        ws, ws_bytes = aggregate_split_planes(x, adj, n_out, plan)
        gi_buf = torch.empty(rows * n_out, dtype=torch.float32, device=x.device) if plan is not None else _gi_buffer(n, K, hid, x.device)
        with _timed("linear_split", rows=rows, k=d, n_out=n_out, presplit=True):
            if plane_cache_enabled():          # W_ih's packed operand: built once per weight version, not per call
                wp = _plane_cache.packed(rnn.weight_ih_l0, lib)
                check(lib.ctgcn_linear_packed_f32(rows, n_out, d, ptr(ws), ptr(wp), ptr(bias), _lib.ACT_NONE, ptr(gi_buf), n_out, _stream()),
                      "ctgcn_linear_packed_f32")
            else:
                check(lib.ctgcn_linear_presplit_f32(rows, n_out, d, ptr(w_ih), w_ih.stride(0), ptr(bias), ptr(gi_buf), n_out, ptr(ws), ws_bytes,
                                                    _stream()), "ctgcn_linear_presplit_f32")
        with _timed("gru_seq", rows=n, steps=K):
            check(lib.ctgcn_gru_seq_f32(n, K, hid, ptr(gi_buf), ptr(w_hh), ptr(b_hn), ptr(ln_w), ptr(ln_b), eps, 1, ptr(out), out.stride(0),
                                        None, forward_split_mode(), 0, ptr(plan["order"]) if plan is not None else None,
                                        ptr(plan["tile_mask"]) if plan is not None else None, ptr(plan["tile_base"]) if plan is not None else None,
                                        _stream()), "ctgcn_gru_seq_f32")
    return out


# ------------------------------------------------------------- CoreDiffusion layer, training form (d_in = hidden = 128, row plan)
def train_fused_enabled():
    """CTGCN_TRAIN_FUSED=0: training keeps round 3's path (fp32 H, _CoreAggregate + _GruSeq) for A/B runs."""
    import os
    return os.environ.get("CTGCN_TRAIN_FUSED", "1") != "0"


def core_diffusion_fused_ok(rnn, norm, x, adj):
    """Training through a CoreDiffusion layer (layers.py:41-62) with d_in = hidden = 128: the forward is the inference path (planes + row
    plan, no fp32 H), the backward runs ctgcn_gru_bwd_rec_f32 / ctgcn_gru_bwd_in_f32 (see _CoreDiffusionFused).  The plan's backward
    relies on a row having no entry tagged with a slot it repeats, seen from the transposed side too: symmetric lists only."""
    if not train_fused_enabled() or not aggregate_split_enabled() or not layer_kernel_enabled(True) or forward_split_mode() != 2:
        return False
    if x.dim() != 2 or not gru_fused_ok(rnn, x) or x.shape[0] != adj.n or x.device != adj.device or not isinstance(norm, torch.nn.LayerNorm):
        return False
    if not norm.elementwise_affine or norm.bias is None:
        return False
    d, hid = x.shape[1], rnn.hidden_size
    if d != hid or rnn.input_size != hid or x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16 or adj.K < 1 or adj.K > 64 or adj.n < 1:
        return False              # (33-64 matrices — America-Air / Europe-Air depth — since round 6: two mask words per tile in the backward kernels too)
    return bool(adj.symmetric) and adj.n * adj.K * (d * 4 + 4) <= _AGG_SPLIT_MAX and rnn.weight_ih_l0.is_contiguous()


def _u32ptr(t, offset=0):
    return None if t is None else t.data_ptr() + 4 * int(offset)


_kept_planes = {"bytes": 0, "budget": int(float(_os.environ.get("CTGCN_TRAIN_PLANES_GB", "32")) * (1 << 30))}


def _release_planes(nbytes):
    _kept_planes["bytes"] -= nbytes


class _CoreDiffusionFused(torch.autograd.Function):
    """out = LayerNorm(sum_k GRU(relu(cumulative A_k x))_k) with autograd, d_in = hidden = 128.
    forward  = ctgcn_core_aggregate_split_f32 (fp16 planes, row plan) + ctgcn_gru_layer_presplit_f32: exactly the inference path; the planes
               are what is kept for the backward (the rows the plan skips are never written: 47 % of them on config 5).
    backward = per chunk of positions: recompute (gates, h, pre-norm sum) -> LayerNorm backward -> backward recurrence + dW_hh ->
               dx + dW_ih + the aggregation's mask / suffix sums (Z, S0); then ONE gather (ctgcn_core_aggregate_bwd_f32)."""

    @staticmethod
    def forward(ctx, x, adj, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b, eps):
        lib = _lib.load()
        n, K, hid = adj.n, adj.K, w_hh.shape[1]
        plan = adj.row_plan() if row_plan_enabled() else None
        x_d = x.detach()
        with torch.cuda.device(x.device):
            ws, _ = aggregate_split_planes(x_d, adj, 1, plan)
            bias, b_hn = _fold_gru_bias(b_ih, b_hh, hid)
            out = torch.empty(n, hid, dtype=torch.float32, device=x.device)
            with _timed("gru_layer", rows=n, steps=K, reduce_sum=True, presplit=True, new_rows=(plan["new_rows"] if plan is not None else n * K)):
                check(lib.ctgcn_gru_layer_presplit_f32(n, K, hid, ptr(ws), ptr(w_ih.detach()), ptr(w_hh.detach().contiguous()), ptr(bias), ptr(b_hn),
                                                       ptr(ln_w.detach()), ptr(ln_b.detach()), eps, ptr(out), out.stride(0),
                                                       ptr(plan["order"]) if plan is not None else None,
                                                       ptr(plan["tile_mask"]) if plan is not None else None, _stream()), "ctgcn_gru_layer_presplit_f32")
        ctx.adj, ctx.plan, ctx.eps = adj, plan, eps
        # The planes are what the backward reads (recompute pass, dx / dW_ih kernel).  Keeping them for every layer of a config-5 window is
        # 132 GB (226 GB peak of 288); beyond a budget of kept bytes (CTGCN_TRAIN_PLANES_GB, default 32) a layer keeps only its INPUT — which
        # autograd holds anyway as the previous layer's output — and writes the planes again in its backward: the same kernel on the same
        # input, bit-identical gradients, +1 ms per 1 M-node layer.
        nbytes = ws.numel()
        keep = _kept_planes["bytes"] + nbytes <= _kept_planes["budget"]
        ctx.kept = keep
        if keep:                                          # accounted for as long as the planes live (freed by the backward, or with a dropped graph)
            import weakref
            _kept_planes["bytes"] += nbytes
            weakref.finalize(ws, _release_planes, nbytes)
        ctx.save_for_backward(ws if keep else x_d, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        ws, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b = ctx.saved_tensors
        if not ctx.kept:
            with torch.cuda.device(ws.device):
                ws, _ = aggregate_split_planes(ws, ctx.adj, 1, ctx.plan)          # ws was the layer's input x
        adj, plan, eps = ctx.adj, ctx.plan, ctx.eps
        n, K, hid = adj.n, adj.K, w_hh.shape[1]
        dev = ws.device
        w_ih_d, w_hh_d = w_ih.detach(), w_hh.detach().contiguous()
        bias, b_hn = _fold_gru_bias(b_ih, b_hh, hid)
        if not (dout.dim() == 2 and dout.stride(1) == 1 and dout.stride(0) >= hid and dout.stride(0) % 2 == 0 and dout.data_ptr() % 8 == 0):
            dout = dout.contiguous()       # a column of a [n, T, 128] gradient is read in place
        order = plan["order"] if plan is not None else None
        tmask = plan["tile_mask"] if plan is not None else None
        chunks = _row_chunks(lib, n, K, hid)
        cmax = chunks[0][1]
        nb = int(lib.ctgcn_gru_bwd_blocks(cmax))
        with torch.cuda.device(dev):
            gates = torch.empty(cmax * K, 3 * hid, dtype=torch.float32, device=dev)      # r, z, q: n is rebuilt in the backward kernel
            hseq = torch.empty(cmax * K, hid, dtype=torch.float32, device=dev)
            presum = torch.empty(cmax, hid, dtype=torch.float32, device=dev)
            dpre = torch.empty(cmax, hid, dtype=torch.float32, device=dev)
            dgi = torch.empty(cmax * K, 3 * hid, dtype=torch.float32, device=dev)
            dw_hh_part = torch.zeros(nb, 3 * hid, hid, dtype=torch.float32, device=dev)
            dw_ih_part = torch.zeros(nb, 3 * hid, hid, dtype=torch.float32, device=dev)
            dbn_part = torch.zeros(nb, hid, dtype=torch.float32, device=dev)
            dbi_part = torch.zeros(nb, 3 * hid, dtype=torch.float32, device=dev)
            ln_part = torch.empty(2048, 2 * hid, dtype=torch.float32, device=dev)
            ln_sum = torch.zeros(2 * hid, dtype=torch.float32, device=dev)
            Z = torch.empty(n, K, hid, dtype=torch.float32, device=dev)
            S0 = torch.empty(n, hid, dtype=torch.float32, device=dev) if adj.self_loop else None
            for lo, cnt in chunks:
                tm = _u32ptr(tmask, (lo // 16) * (2 if K > 32 else 1))      # one mask word per 16-row tile, two beyond 32 steps
                od = _u32ptr(order, lo)
                fresh = (plan["new_rows"] / float(n * K)) if plan is not None else 1.0     # fraction of (position, step) rows that bring a new x
                with _timed("gru_layer", rows=cnt, steps=K, reduce_sum=True, presplit=True, save=True, new_rows=int(cnt * K * fresh)):
                    check(lib.ctgcn_gru_layer_presplit_save_f32(cnt, K, hid, ptr(ws), n * K, lo, ptr(w_ih_d), ptr(w_hh_d), ptr(bias), ptr(b_hn), tm,
                                                                ptr(gates), ptr(hseq), ptr(presum), _stream()), "ctgcn_gru_layer_presplit_save_f32")
                dy = dout if order is not None else dout[lo:lo + cnt]
                check(lib.ctgcn_layernorm_bwd_f32(cnt, 1, hid, ptr(presum), ptr(dy), dy.stride(0), ptr(ln_w.detach()), eps, ptr(dpre), ptr(ln_part),
                                                  ln_part.shape[0], od, _stream()), "ctgcn_layernorm_bwd_f32")
                ln_sum += ln_part.sum(0)
                with _timed("gru_bwd_rec", rows=cnt, steps=K, fresh=fresh, per_step=False):
                    check(lib.ctgcn_gru_bwd_rec_f32(cnt, K, hid, ptr(gates), 3, ptr(hseq), ptr(dpre), None, ptr(w_hh_d), tm, ptr(dgi), ptr(dw_hh_part),
                                                    ptr(dbn_part), nb, 1, _stream()), "ctgcn_gru_bwd_rec_f32")
                with _timed("gru_bwd_in", rows=cnt, steps=K, fresh=fresh, z_out=True):
                    check(lib.ctgcn_gru_bwd_in_f32(cnt, K, hid, ptr(dgi), ptr(w_ih_d), tm, ptr(ws), n * K, lo, None, 0, None,
                                                   ptr(Z) if order is not None else ptr(Z[lo:]),
                                                   ptr(S0) if (S0 is None or order is not None) else ptr(S0[lo:]), od, 1 if adj.nested else 0,
                                                   ptr(dw_ih_part), ptr(dbi_part), nb, 1, _stream()), "ctgcn_gru_bwd_in_f32")
            dX = _aggregate_gather(adj, Z, S0, True) if ctx.needs_input_grad[0] else None
        dw_ih = dw_ih_part.sum(0)
        dw_hh = dw_hh_part.sum(0)
        db_ih = db_hh = None
        if b_ih is not None:
            db_ih = dbi_part.sum(0)
            db_hh = torch.cat([db_ih[: 2 * hid], dbn_part.sum(0)])
        return dX, None, dw_ih, dw_hh, db_ih, db_hh, ln_sum[:hid].clone(), ln_sum[hid:].clone(), None


def _fold_gru_bias(b_ih, b_hh, hid):
    """(b_ih + b_hh on the r and z gates | b_ih on n,  b_hn) — the two bias vectors the GRU kernels take; (None, None) without bias"""
    if b_ih is None:
        return None, None
    bias = b_ih.detach().clone()
    bias[: 2 * hid] += b_hh.detach()[: 2 * hid]
    return bias, b_hh.detach()[2 * hid:].contiguous()


def core_diffusion_fused(x, adj, rnn, norm):
    """CoreDiffusion.forward (layers.py:41-62) with autograd, see _CoreDiffusionFused (core_diffusion_fused_ok decides)."""
    b_ih = rnn.bias_ih_l0 if rnn.bias else None
    b_hh = rnn.bias_hh_l0 if rnn.bias else None
    return _CoreDiffusionFused.apply(x, adj, rnn.weight_ih_l0, rnn.weight_hh_l0, b_ih, b_hh, norm.weight, norm.bias, float(norm.eps))


# ------------------------------------------------------------- a window of small snapshots: one launch per kernel
_GROUP_MAX_NODES = 200_000


def _compute_units(dev):
    """CUs the persistent kernels of `dev` run on — queried under THAT device (the gate asked the current device, the C side the
    tensors' device at launch; on a process with several GPUs the two could disagree and the launch refused what the gate had accepted)"""
    with torch.cuda.device(dev):
        return int(_lib.load().ctgcn_compute_units())


class _GroupTables(object):
    """(device table, host shadow) pairs of the grouped launches (include/ctgcn_hip.h, ABI 28), one per call site, window length and stream,
    kept between forwards: when a call's descriptors equal the shadow the C side writes nothing — the steady state of an inference loop,
    where the caching allocator hands the same buffers to the same places forward after forward.  The comparison is on the descriptor
    bytes, so an entry can never make a launch read a stale table; entries are small (a few KB) and least-recently-used ones go first.
    Under hipGraph capture nothing is cached: a fresh table from the capture's pool, written by the (capturable) writer kernel."""

    SLAB_BYTES = 8 << 20

    def __init__(self, limit=1024):
        import collections
        self.limit, self.entries = limit, collections.OrderedDict()
        self.slabs = {}                       # device -> [slab tensor, bytes used]: tables are carved from ONE allocation per device
        self.misses = {}                      # call site -> descriptor sets seen for the first time (diagnostic)

    def _carve(self, dev, nbytes):
        """256-byte aligned device memory for a table.  Not torch.empty per table: a table allocated on a miss changes the caching allocator's
        small pool, the next forward's bias / output tensors land elsewhere, their addresses are in the descriptors — another miss, another
        allocation: the cache never converged (measured: 4 of 6 tables rewritten per forward, for ever).  A full slab drops every entry of
        its device and starts over."""
        key = str(dev)
        need = -(-nbytes // 256) * 256
        slab = self.slabs.get(key)
        if slab is None or slab[1] + need > slab[0].numel():
            if slab is not None:
                for k in [k for k in self.entries if k[2] == key]:
                    del self.entries[k]
                self._retire(slab[0])
            slab = self.slabs[key] = [torch.empty(max(self.SLAB_BYTES, need), dtype=torch.uint8, device=dev), 0]
        view = slab[0][slab[1]: slab[1] + need]
        slab[1] += need
        return view

    def get(self, lib, site, groups, dev, desc=b""):
        """desc: the caller's descriptor bytes (its ctypes arrays).  They are part of the key: the caching allocator may alternate between two
        or three sets of addresses (a forward runs while the previous forward's output is still alive), and every set keeps its own table."""
        nbytes = int(lib.ctgcn_group_table_bytes(groups))
        if torch.cuda.is_current_stream_capturing():
            return torch.empty(nbytes, dtype=torch.uint8, device=dev), None, nbytes
        key = (site, groups, str(dev), torch.cuda.current_stream(dev).cuda_stream, hash(desc))
        e = self.entries.get(key)
        if e is None:
            self.misses[site] = self.misses.get(site, 0) + 1
            if len(self.entries) >= self.limit:            # (carved memory is not returned piecemeal: start over)
                self.entries.clear()
                for old in self.slabs.values():
                    self._retire(old[0])
                self.slabs.clear()
            e = self.entries[key] = (self._carve(dev, nbytes), ctypes.create_string_buffer(nbytes))
        else:
            self.entries.move_to_end(key)
        return e[0], e[1], nbytes

    def _retire(self, slab):
        """a slab leaves the cache: kernels queued on ANY stream may still read tables in it, so the device is drained before the caching
        allocator gets the block back (rare: a slab holds ~500 tables)"""
        torch.cuda.synchronize(slab.device)

    def clear(self):
        for old in self.slabs.values():
            self._retire(old[0])
        self.entries.clear()
        self.slabs.clear()


_group_tables = _GroupTables()


def group_launch_enabled():
    """CTGCN_GROUP=0: every snapshot launches its own kernels (round 3's path: four HIP streams overlap the tails) for A/B runs."""
    import os
    return os.environ.get("CTGCN_GROUP", "1") != "0"        # (capturable since ABI 28: the descriptor tables travel as kernel arguments)


def core_diffusion_group_ok(xs, adjs, rnns, norms):
    """The width-128 CoreDiffusion layer of a window's snapshots in ONE aggregation launch + ONE GRU layer launch
    (ctgcn_core_aggregate_split_group_f32 / ctgcn_gru_layer_presplit_group_f32): inference, >= 2 snapshots of a small graph that share
    the node set, no hub rows, every snapshot fit for the per-snapshot split path."""
    if not group_launch_enabled() or len(xs) < 2 or not xs[0].is_cuda or len(xs) > _compute_units(xs[0].device):
        return False          # the grouped GRU launch gives every snapshot at least one block of the persistent grid: longer windows take the per-snapshot path
    n = adjs[0].n
    if n > _GROUP_MAX_NODES:
        return False
    planned = None
    for x, adj, rnn, norm in zip(xs, adjs, rnns, norms):
        if adj.n != n or x.dim() != 2 or x.shape[1] != 128 or rnn.hidden_size != 128 or not aggregate_split_ok(rnn, x, adj):
            return False
        if adj.K > 32 or adj.long_rows() is not None or not isinstance(norm, torch.nn.LayerNorm) or norm.weight is None or norm.bias is None:
            return False
        p = (adj.row_plan() is not None) if row_plan_enabled() else False
        if planned is None:
            planned = p
        if p != planned:
            return False
    return True


def core_diffusion_split_group(xs, adjs, rnns, norms, outs):
    """outs[t] = LayerNorm(sum_k GRU_t(relu(cumulative A_{t,k} xs[t]))_k) for every snapshot t of the window — core_diffusion_split(d = 128)
    per snapshot, as two launches for the whole window.  Bit-identical to the per-snapshot calls (same kernels' code per snapshot)."""
    lib = _lib.load()
    T = len(xs)
    n = adjs[0].n
    dev = xs[0].device
    agg = (_lib.AggSplitGroup * T)()
    lay = (_lib.GruLayerGroup * T)()
    keep = []                                         # tensors the descriptors point at, alive until the launches are queued
    use_plan = row_plan_enabled()
    with torch.cuda.device(dev):
        for t, (x, adj, rnn, norm, out) in enumerate(zip(xs, adjs, rnns, norms, outs)):
            K = adj.K
            plan = adj.row_plan() if use_plan else None
            ws_bytes = int(lib.ctgcn_core_aggregate_split_workspace_bytes(n, 128, K, 1, 0))
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            bias, b_hn = _gru_bias(rnn, 128)
            w_ih, w_hh = rnn.weight_ih_l0.detach(), rnn.weight_hh_l0.detach().contiguous()
            keep.append((ws, bias, b_hn, w_hh, x))
            a = agg[t]
            a.row_ptr, a.col_idx, a.val, a.slot = ptr(adj.row_ptr), ptr(adj.col), ptr(adj.val), ptr(adj.slot)
            a.X, a.ldx, a.K, a.flags = ptr(x), x.stride(0), K, adj.flags | _lib.F_RELU
            a.row_order = ptr(plan["order"]) if plan is not None else None
            a.tile_mask = ptr(plan["tile_mask"]) if plan is not None else None
            a.workspace, a.workspace_bytes = ptr(ws), ws_bytes
            g = lay[t]
            g.planes, g.w_ih, g.w_hh, g.bias_gi, g.b_hn = ptr(ws), ptr(w_ih), ptr(w_hh), ptr(bias), ptr(b_hn)
            g.ln_weight, g.ln_bias, g.ln_eps, g.steps = ptr(norm.weight), ptr(norm.bias), float(norm.eps), K
            g.out, g.ld_out = ptr(out), out.stride(0)
            g.row_order, g.tile_mask = a.row_order, a.tile_mask
            g.work = (plan["new_rows"] if plan is not None else n * K) + n * K
        tb_agg, sh_agg, tb_bytes = _group_tables.get(lib, "agg128", T, dev, bytes(agg))
        tb_lay, sh_lay, _ = _group_tables.get(lib, "layer128", T, dev, bytes(lay))
        nnz = sum(adj.nnz for adj in adjs)
        rows_written = sum((adj.row_plan()["new_rows"] if use_plan else n * adj.K) for adj in adjs)
        with _timed("agg_fwd", n=n, d=128, K=max(adj.K for adj in adjs), nnz=nnz, split=True, group=T, rows_written=rows_written,
                    K_sum=sum(adj.K for adj in adjs)):
            check(lib.ctgcn_core_aggregate_split_group_f32(T, n, 128, agg, ptr(tb_agg), tb_bytes, sh_agg, _stream()), "ctgcn_core_aggregate_split_group_f32")
        with _timed("gru_layer", rows=n, steps=max(adj.K for adj in adjs), reduce_sum=True, presplit=True, group=T, new_rows=rows_written,
                    row_steps=sum(n * adj.K for adj in adjs)):
            check(lib.ctgcn_gru_layer_presplit_group_f32(T, n, 128, lay, ptr(tb_lay), tb_bytes, sh_lay, _stream()), "ctgcn_gru_layer_presplit_group_f32")
    del keep
    return outs


def core_diffusion_wide_group_ok(xs, adjs, rnns, norms):
    """The first CoreDiffusion layer (d_in != 128: the 500-wide GRU input of the 'C' configs, layers.py:59) of a window's snapshots in one
    launch per kernel — aggregation into operand planes shared by the window, ONE panel GEMM over all snapshots' rows, one recurrence launch:
    the conditions of core_diffusion_group_ok with the split GEMM (packed weights) as consumer."""
    if not group_launch_enabled() or len(xs) < 2 or not xs[0].is_cuda or len(xs) > _compute_units(xs[0].device):
        return False
    if forward_split_mode() != 2 or not plane_cache_enabled():
        return False
    n, d = adjs[0].n, xs[0].shape[1] if xs[0].dim() == 2 else 0
    if n > _GROUP_MAX_NODES or d == 128:
        return False
    planned = None
    for x, adj, rnn, norm in zip(xs, adjs, rnns, norms):
        if adj.n != n or x.dim() != 2 or x.shape[1] != d or rnn.hidden_size != 128 or not aggregate_split_ok(rnn, x, adj):
            return False
        if adj.K > 32 or adj.long_rows() is not None or not isinstance(norm, torch.nn.LayerNorm) or norm.weight is None or norm.bias is None:
            return False
        p = (adj.row_plan(adj.PLAN_TILE_GEMM) is not None) if row_plan_enabled() else False
        if planned is None:
            planned = p
        if p != planned:
            return False
    return True


_panel_group_cache = {}


def _panel_groups(padded, dev):
    """int32 [sum(padded) / 128] on the device: the group of every 128-row panel of the shared operand planes (a function of the graphs'
    row counts only: built once per window shape)"""
    key = (str(dev), tuple(padded))
    hit = _panel_group_cache.get(key)
    if hit is None:
        if len(_panel_group_cache) > 64:
            _panel_group_cache.clear()
        counts = torch.tensor([p // 128 for p in padded], dtype=torch.int64)
        hit = _panel_group_cache[key] = torch.repeat_interleave(torch.arange(len(padded), dtype=torch.int32), counts).to(dev)
    return hit


def core_diffusion_wide_group(xs, adjs, rnns, norms, outs):
    """outs[t] = LayerNorm(sum_k GRU_t(relu(cumulative A_{t,k} xs[t]))_k) for every snapshot t — core_diffusion_split(d_in != 128) per snapshot,
    as three launches for the whole window (reference models.py:243-247 loops over the snapshots).  The operand rows of all snapshots (compact
    under the row plans, every snapshot padded to whole 128-row panels) share one pair of planes and one gi buffer.  Bit-identical to the
    per-snapshot calls: the same kernels' code per row."""
    lib = _lib.load()
    T = len(xs)
    n, d = xs[0].shape
    dev = xs[0].device
    hid, n_out = 128, 384
    kp = -(-d // 64) * 64
    use_plan = row_plan_enabled()
    plans = [adj.row_plan(adj.PLAN_TILE_GEMM) if use_plan else None for adj in adjs]
    rows = [(pl["operand_rows"] if pl is not None else n * adj.K) for pl, adj in zip(plans, adjs)]
    padded = [-(-r // 128) * 128 for r in rows]
    first = [0] * T
    for t in range(1, T):
        first[t] = first[t - 1] + padded[t - 1]
    total = first[-1] + padded[-1]
    agg = (_lib.AggSplitGroup * T)()
    seq = (_lib.GruSeqGroup * T)()
    w_arr, b_arr = (ctypes.c_void_p * T)(), (ctypes.c_void_p * T)()
    keep = []
    with torch.cuda.device(dev):
        p1 = torch.empty(total * kp, dtype=torch.float16, device=dev)
        p2 = torch.empty(total * kp, dtype=torch.float16, device=dev)
        sc = torch.empty(total, dtype=torch.float32, device=dev)
        gi = torch.empty(total, n_out, dtype=torch.float32, device=dev)
        pg = _panel_groups(padded, dev)
        for t, (x, adj, rnn, norm, out, plan) in enumerate(zip(xs, adjs, rnns, norms, outs, plans)):
            K = adj.K
            bias, b_hn = _gru_bias(rnn, hid)
            w_hh = rnn.weight_hh_l0.detach().contiguous()
            wp = _plane_cache.packed(rnn.weight_ih_l0, lib)
            keep.append((bias, b_hn, w_hh, wp, x))
            a = agg[t]
            a.row_ptr, a.col_idx, a.val, a.slot = ptr(adj.row_ptr), ptr(adj.col), ptr(adj.val), ptr(adj.slot)
            a.X, a.ldx, a.K, a.flags = ptr(x), x.stride(0), K, adj.flags | _lib.F_RELU
            a.row_order = ptr(plan["order"]) if plan is not None else None
            a.tile_mask = ptr(plan["tile_mask"]) if plan is not None else None
            a.tile_base = ptr(plan["tile_base"]) if plan is not None else None
            a.workspace, a.workspace_bytes = None, 0
            a.planes1 = ctypes.c_void_p(p1.data_ptr() + first[t] * kp * 2)
            a.planes2 = ctypes.c_void_p(p2.data_ptr() + first[t] * kp * 2)
            a.scales = ctypes.c_void_p(sc.data_ptr() + first[t] * 4)
            w_arr[t], b_arr[t] = wp.data_ptr(), (bias.data_ptr() if bias is not None else None)
            g = seq[t]
            g.gi, g.w_hh, g.b_hn = ctypes.c_void_p(gi.data_ptr() + first[t] * n_out * 4), ptr(w_hh), ptr(b_hn)
            g.ln_weight, g.ln_bias, g.ln_eps, g.steps = ptr(norm.weight), ptr(norm.bias), float(norm.eps), K
            g.out, g.ld_out = ptr(out), out.stride(0)
            g.row_order, g.tile_mask, g.tile_base = a.row_order, a.tile_mask, a.tile_base
            g.work = n * K + rows[t]
        tb_agg, sh_agg, tb_bytes = _group_tables.get(lib, "aggwide", T, dev, bytes(agg))
        tb_lin, sh_lin, _ = _group_tables.get(lib, "linwide", T, dev, bytes(w_arr) + bytes(b_arr))
        tb_seq, sh_seq, _ = _group_tables.get(lib, "seqwide", T, dev, bytes(seq))
        nnz = sum(adj.nnz for adj in adjs)
        Kmax = max(adj.K for adj in adjs)
        with _timed("agg_fwd", n=n, d=d, K=Kmax, nnz=nnz, split=True, group=T, rows_written=sum(rows), K_sum=sum(adj.K for adj in adjs)):
            check(lib.ctgcn_core_aggregate_split_group_f32(T, n, d, agg, ptr(tb_agg), tb_bytes, sh_agg, _stream()), "ctgcn_core_aggregate_split_group_f32")
        with _timed("linear_split", rows=total, k=d, n_out=n_out, presplit=True, group=T):
            check(lib.ctgcn_linear_packed_group_f32(T, total, n_out, d, ptr(p1), ptr(p2), ptr(sc), ptr(pg), w_arr, b_arr, _lib.ACT_NONE, ptr(gi), n_out,
                                                    ptr(tb_lin), tb_bytes, sh_lin, _stream()), "ctgcn_linear_packed_group_f32")
        with _timed("gru_seq", rows=n, steps=Kmax, group=T, row_steps=sum(n * adj.K for adj in adjs)):
            check(lib.ctgcn_gru_seq_group_f32(T, n, hid, seq, ptr(tb_seq), tb_bytes, sh_seq, _stream()), "ctgcn_gru_seq_group_f32")
    del keep
    return outs


def linear_of_identity_group(weights, biases):
    """[W_t^T + b_t for t] — nn.Linear applied to one-hot features (helper.py:161-172 with x = I, layers.py:95-106) for every snapshot of a
    window in ONE transpose launch (inference; weights [d, n] fp32 of one shape and stride)."""
    lib = _lib.load()
    T = len(weights)
    d, n = weights[0].shape
    dev = weights[0].device
    ws = [w.detach() for w in weights]
    outs = [torch.empty(n, d, dtype=torch.float32, device=dev) for _ in range(T)]
    w_arr, o_arr = (ctypes.c_void_p * T)(*[w.data_ptr() for w in ws]), (ctypes.c_void_p * T)(*[o.data_ptr() for o in outs])
    has_bias = all(b is not None for b in biases)
    b_arr = (ctypes.c_void_p * T)(*[b.detach().data_ptr() for b in biases]) if has_bias else None
    with torch.cuda.device(dev):
        table, shadow, tb_bytes = _group_tables.get(lib, "transpose", T, dev, bytes(w_arr) + bytes(o_arr) + (bytes(b_arr) if has_bias else b""))
        with _timed("transpose_bias", n=n, d=d, group=T):
            check(lib.ctgcn_transpose_bias_group_f32(T, n, d, w_arr, ws[0].stride(0), b_arr, o_arr, d, ptr(table), tb_bytes, shadow, _stream()),
                  "ctgcn_transpose_bias_group_f32")
    return outs


def linear_of_identity_group_ok(weights, biases):
    if not group_launch_enabled() or len(weights) < 2 or len(weights) > 1024:
        return False
    w0 = weights[0]
    none = biases[0] is None
    for w, b in zip(weights, biases):
        if not (w.is_cuda and w.dtype == torch.float32 and w.dim() == 2 and w.shape == w0.shape and w.stride() == w0.stride() and w.stride(1) == 1
                and w.device == w0.device):
            return False
        if (b is None) != none or (b is not None and not (b.is_cuda and b.dtype == torch.float32 and b.is_contiguous())):
            return False
    return True


def _accumulate_tn(out, a2d, b2d):
    """out[M,N] += a2d[R,M]^T @ b2d[R,N] for R >> M,N (weight gradients: R = rows*steps).  A plain TN GEMM with a
    384x128 output only fills a few dozen workgroups; splitting R into S batches (strided batched GEMM, no copies)
    and summing the S partial products keeps all CUs busy."""
    R, M = a2d.shape
    S = 1
    while S < 256 and R % (2 * S) == 0 and R // (2 * S) >= 1024:
        S *= 2
    if S == 1:
        out.addmm_(a2d.t(), b2d)
        return
    part = torch.bmm(a2d.view(S, R // S, M).transpose(1, 2), b2d.view(S, R // S, b2d.shape[1]))
    out += part.sum(0)


def _zero_rnn_grads(seq, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b):
    """the nine gradients of _GruSeq / _LstmSeq for an input without rows"""
    z = lambda t: None if t is None else torch.zeros_like(t)
    return torch.zeros_like(seq), z(w_ih), z(w_hh), z(b_ih), z(b_hh), z(ln_w), z(ln_b), None, None


class _GruSeq(torch.autograd.Function):
    """LayerNorm(sum_t GRU(seq)_t) or LayerNorm(GRU(seq)).  Forward = the fused inference kernels.  Backward
    recomputes the raw h sequence and the gates chunk by chunk (so nothing but `seq` is kept alive between forward
    and backward), runs the HIP backward recurrence and leaves the weight/input GEMMs to hipBLASLt."""

    @staticmethod
    def forward(ctx, seq, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b, eps, reduce_sum):
        hid = w_hh.shape[1]
        if b_ih is not None:
            bias = b_ih.detach().clone()
            bias[: 2 * hid] += b_hh.detach()[: 2 * hid]
            b_hn = b_hh.detach()[2 * hid:].contiguous()
        else:
            bias, b_hn = None, None
        seq_c = seq.contiguous()
        # d_in != hidden (the 500-wide first layer): the projection gi of a call that runs in one chunk is kept for the backward when it fits
        # the budget of kept training buffers (CTGCN_TRAIN_PLANES_GB, shared with _CoreDiffusionFused's planes) — the recompute pass then
        # skips the split of x and the GEMM (Enron-like: 0.7 of ~6 ms per snapshot).  The same bits: it IS the forward's gi.
        kept = [] if (seq_c.shape[2] != hid and keep_projection_enabled()
                      and _kept_planes["bytes"] + seq_c.shape[0] * seq_c.shape[1] * 3 * hid * 4 <= _kept_planes["budget"]) else None
        out = _gru_forward(seq_c, w_ih.detach(), w_hh.detach().contiguous(), bias, b_hn, ln_w, ln_b, eps, reduce_sum, keep_gi=kept)
        ctx.save_for_backward(seq_c, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b)
        ctx.eps, ctx.reduce_sum = eps, reduce_sum
        ctx.kept_gi = None
        if kept:
            import weakref
            gi_buf, blocked = kept[0]
            nbytes = gi_buf.numel() * 4
            _kept_planes["bytes"] += nbytes
            weakref.finalize(gi_buf, _release_planes, nbytes)
            ctx.kept_gi = (gi_buf, blocked)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        seq, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b = ctx.saved_tensors
        reduce_sum, eps = ctx.reduce_sum, ctx.eps
        rows, steps, d_in = seq.shape
        hid = w_hh.shape[1]
        dev = seq.device
        if rows == 0:                     # an empty node slice (a snapshot-parallel rank that owns nothing): all gradients are zero
            return _zero_rnn_grads(seq, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b)
        w_ih_d, w_hh_d = w_ih.detach(), w_hh.detach().contiguous()
        if b_ih is not None:
            bias = b_ih.detach().clone()
            bias[: 2 * hid] += b_hh.detach()[: 2 * hid]
            b_hn = b_hh.detach()[2 * hid:].contiguous()
        else:
            bias, b_hn = None, None
        # a [rows, 128] gradient that is a column of a [rows, T, 128] tensor (the last CoreDiffusion of a snapshot under the temporal GRU)
        # is read in place by the LayerNorm backward kernel; anything else is made dense
        strided_ok = (ln_w is not None and reduce_sum and dout.dim() == 2 and dout.stride(1) == 1 and dout.stride(0) >= hid
                      and dout.stride(0) % 2 == 0 and dout.data_ptr() % 8 == 0)
        if not strided_ok:
            dout = dout.contiguous()
        dseq = torch.empty_like(seq)
        dw_ih = torch.zeros_like(w_ih_d)
        dw_hh = torch.zeros_like(w_hh_d)
        db_all = torch.zeros(4 * hid, dtype=torch.float32, device=dev)                      # [d_gi (3h) | d_ghn (h)] column sums
        bias_part = torch.empty(512, 4 * hid, dtype=torch.float32, device=dev)
        dln_w = torch.zeros(hid, dtype=torch.float32, device=dev) if ln_w is not None else None
        dln_b = torch.zeros(hid, dtype=torch.float32, device=dev) if ln_w is not None else None
        ln_part = torch.empty(2048, 2 * hid, dtype=torch.float32, device=dev) if ln_w is not None else None
        chunks = _row_chunks(lib, rows, steps, hid)
        cmax = chunks[0][1]
        kept_gi, ctx.kept_gi = ctx.kept_gi, None          # used once: the buffer becomes d_gi below (a second backward recomputes)
        if kept_gi is not None and len(chunks) != 1:
            kept_gi = None
        gi_flat = kept_gi[0] if kept_gi is not None else _gi_buffer(cmax, steps, hid, dev)      # reused as d_gi
        gates_buf = torch.empty(cmax * steps, 4 * hid, dtype=torch.float32, device=dev)
        hseq_buf = torch.empty(cmax, steps, hid, dtype=torch.float32, device=dev)
        dghn_buf = torch.empty(cmax * steps, hid, dtype=torch.float32, device=dev)
        split = split_mfma_enabled()
        # d_in = hidden = 128: the two resident-weight backward kernels of ctgcn_gru_bwd.hip (backward recurrence + dW_hh; dx + dW_ih) instead
        # of gru_seq_bwd_x3 + gru_dx_x3 + 2 x gru_dw_x3: d_gi is the only intermediate, dGH / dGHn never leave the CU
        fused_bwd = split and train_fused_enabled() and forward_split_mode() == 2 and d_in == hid and steps <= 64 and w_ih_d.is_contiguous()
        if fused_bwd:
            nb = int(lib.ctgcn_gru_bwd_blocks(cmax))
            dw_part_ih = torch.zeros(nb, 3 * hid, hid, dtype=torch.float32, device=dev)
            dw_part_hh = torch.zeros(nb, 3 * hid, hid, dtype=torch.float32, device=dev)
            dbn_part = torch.zeros(nb, hid, dtype=torch.float32, device=dev)
            dbi_part = torch.zeros(nb, 3 * hid, dtype=torch.float32, device=dev)
        elif split:
            dw_part_ih = torch.empty(_DW_PAIRS, 3 * hid, hid, dtype=torch.float32, device=dev)    # summed once at the end
            dw_part_hh = torch.empty(_DW_PAIRS, 3 * hid, hid, dtype=torch.float32, device=dev)
        wide_dw = split and not fused_bwd and hid < d_in <= 1024 and wide_weight_grad_enabled() and seq.stride(2) == 1
        if wide_dw:
            wide_cols = [min(c0, d_in - hid) for c0 in range(0, d_in, hid)]
            dw_part_wide = [torch.empty(_DW_PAIRS, 3 * hid, hid, dtype=torch.float32, device=dev) for _ in wide_cols]
        if not split:                                # read only by the library-GEMM weight gradients below (it was allocated and
            hprev_buf = torch.zeros(cmax, steps, hid, dtype=torch.float32, device=dev)     # zeroed on every split-path backward); [:, 0] stays 0
        with torch.cuda.device(dev):
            for lo, n in chunks:
                x2d = seq[lo:lo + n].reshape(n * steps, d_in)
                gates, hseq = gates_buf[: n * steps], hseq_buf[:n]
                gi = gi_flat[: n * steps * 3 * hid].view(n * steps, 3 * hid)                 # the [rows, 3h] view (d_gi later)
                xs = seq[lo:lo + n]
                if forward_split_mode() == 2 and d_in == hid and layer_kernel_enabled(False) and xs.stride(2) == 1 and xs.stride(1) % 4 == 0 \
                        and xs.stride(0) == steps * xs.stride(1) and xs.data_ptr() % 16 == 0 and w_ih_d.is_contiguous():
                    # recompute in the layer kernel: raw h sequence + gates, the projection consumed from the accumulators (bit-identical
                    # to the pair below, without writing and re-reading gi)
                    with _timed("gru_layer", rows=n, steps=steps, reduce_sum=False, save=True):
                        check(lib.ctgcn_gru_layer_f32(n, steps, d_in, hid, ptr(xs), xs.stride(1), ptr(w_ih_d), ptr(w_hh_d), ptr(bias), ptr(b_hn), None, None,
                                                      0.0, 0, ptr(hseq), 0, ptr(gates), None, 0, _stream()), "ctgcn_gru_layer_f32")
                else:
                    blocked = kept_gi[1] if kept_gi is not None else _project(x2d, w_ih_d, bias, gi_flat, steps_blocked=steps)
                    check(lib.ctgcn_gru_seq_f32(n, steps, hid, ptr(gi_flat), ptr(w_hh_d), ptr(b_hn), None, None, 0.0, 0, ptr(hseq), 0,
                                                ptr(gates), forward_split_mode(), 1 if blocked else 0, None, None, None, _stream()), "ctgcn_gru_seq_f32")
                # LayerNorm backward on the recomputed pre-norm values (dense, tiny next to the recurrence)
                g_out = dout[lo:lo + n]
                if ln_w is not None:
                    # one HIP pass: sum over steps (reduce_sum), mean / rstd recomputed, dpre, per-block partials of d gamma / d beta
                    ln_rows = n if reduce_sum else n * steps
                    dpre = torch.empty((n, hid) if reduce_sum else (n, steps, hid), dtype=torch.float32, device=dev)
                    check(lib.ctgcn_layernorm_bwd_f32(ln_rows, steps if reduce_sum else 1, hid, ptr(hseq), ptr(g_out), g_out.stride(0) if reduce_sum else 0,
                                                      ptr(ln_w.detach()), eps, ptr(dpre), ptr(ln_part), ln_part.shape[0], None, _stream()), "ctgcn_layernorm_bwd_f32")
                    ln_sum = ln_part.sum(0)
                    dln_w += ln_sum[:hid]
                    dln_b += ln_sum[hid:]
                else:
                    dpre = g_out
                dpre = dpre.contiguous()
                dgi, dghn = gi, dghn_buf[: n * steps]                                       # gi is dead: reuse as d_gi
                if fused_bwd:
                    dsq = dseq[lo:lo + n]
                    with _timed("gru_bwd_rec", rows=n, steps=steps, fresh=1.0, per_step=not reduce_sum):
                        check(lib.ctgcn_gru_bwd_rec_f32(n, steps, hid, ptr(gates), 4, ptr(hseq), ptr(dpre) if reduce_sum else None,
                                                        None if reduce_sum else ptr(dpre), ptr(w_hh_d), None, ptr(dgi), ptr(dw_part_hh), ptr(dbn_part),
                                                        nb, 1, _stream()), "ctgcn_gru_bwd_rec_f32")
                    with _timed("gru_bwd_in", rows=n, steps=steps, fresh=1.0, z_out=False):
                        check(lib.ctgcn_gru_bwd_in_f32(n, steps, hid, ptr(dgi), ptr(w_ih_d), None, None, 0, 0, ptr(xs), xs.stride(1), ptr(dsq), None, None,
                                                       None, 0, ptr(dw_part_ih), ptr(dbi_part), nb, 1, _stream()), "ctgcn_gru_bwd_in_f32")
                    continue
                check(lib.ctgcn_gru_seq_bwd_f32(n, steps, hid, ptr(gates), ptr(hseq), None if reduce_sum else ptr(dpre),
                                                ptr(dpre) if reduce_sum else None, ptr(w_hh_d), ptr(dgi), ptr(dghn),
                                                ptr(bias_part), bias_part.shape[0], 1 if split_mfma_enabled() else 0, _stream()),
                      "ctgcn_gru_seq_bwd_f32")
                _project_grad(dgi, w_ih, dseq[lo:lo + n].view(n * steps, d_in))        # the parameter itself: its transpose is cached by identity + version
                db_all += bias_part.sum(0)          # per-block column sums written by the kernel (no re-read of d_gi / d_ghn)
                if split and d_in == hid:
                    _weight_grad(dw_part_ih, dgi, dgi[:, 2 * hid:], x2d, steps, False, lo > 0)
                elif wide_dw:
                    # d_in = 500 (the 'C' configs' first layer): the 128-column weight-gradient kernel on column slices of x (the last one
                    # moved left to end at d_in: its overlap with the slice before is computed twice and stored once) — 4 x 0.26 ms per
                    # Enron-like snapshot against 1.6 ms of the fp32 library TN GEMM (which already runs at 104 TFLOP/s)
                    for part_, c0 in zip(dw_part_wide, wide_cols):
                        _weight_grad(part_, dgi, dgi[:, 2 * hid:], x2d[:, c0:], steps, False, lo > 0)
                else:
                    _accumulate_tn(dw_ih, dgi, x2d)
                if split:                            # h_{t-1} is read from the h sequence with the shift applied in the kernel
                    _weight_grad(dw_part_hh, dgi, dghn, hseq.view(n * steps, hid), steps, True, lo > 0)
                else:
                    hprev = hprev_buf[:n]
                    hprev[:, 1:] = hseq[:, :-1]
                    hp2d = hprev.view(n * steps, hid)
                    _accumulate_tn(dw_hh[: 2 * hid], dgi[:, : 2 * hid], hp2d)
                    _accumulate_tn(dw_hh[2 * hid:], dghn, hp2d)
        if fused_bwd:
            db_all[: 3 * hid] = dbi_part.sum(0)
            db_all[3 * hid:] = dbn_part.sum(0)
        if split:
            dw_hh += dw_part_hh.sum(0)
            if d_in == hid:
                dw_ih += dw_part_ih.sum(0)
            elif wide_dw:
                for part_, c0 in zip(dw_part_wide, wide_cols):
                    dw_ih[:, c0:c0 + hid] = part_.sum(0)
        db_ih = db_hh = None
        if b_ih is not None:
            db_ih = db_all[: 3 * hid].clone()
            db_hh = torch.cat([db_all[: 2 * hid], db_all[3 * hid:]])
        return dseq, dw_ih, dw_hh, db_ih, db_hh, dln_w, (dln_b if ln_b is not None else None), None, None


def lstm_fused_ok(rnn, seq):
    """Fused LSTM recurrence (forward and backward): hidden 128, single layer, batch_first, fp32 CUDA."""
    return (isinstance(rnn, torch.nn.LSTM) and rnn.hidden_size == 128 and rnn.num_layers == 1 and not rnn.bidirectional
            and rnn.batch_first and getattr(rnn, "proj_size", 0) == 0 and seq.is_cuda and seq.dtype == torch.float32)


def _lstm_forward(seq, w_ih, w_hh, bias, ln_w, ln_b, eps, reduce_sum):
    """LayerNorm(sum_t LSTM(seq)_t) / LayerNorm(LSTM(seq)): library GEMM for the input projection, recurrence in ctgcn_lstm_seq_f32."""
    lib = _lib.load()
    rows, steps, d_in = seq.shape
    hid = w_hh.shape[1]
    out = torch.empty((rows, hid) if reduce_sum else (rows, steps, hid), dtype=torch.float32, device=seq.device)
    if rows == 0:
        return out
    chunks = _row_chunks(lib, rows, steps, hid)
    gi_buf = torch.empty(chunks[0][1] * steps, 4 * hid, dtype=torch.float32, device=seq.device)
    with torch.cuda.device(seq.device):
        for lo, n in chunks:
            gi = gi_buf[: n * steps]
            x2d = seq[lo:lo + n].reshape(n * steps, d_in)
            if bias is None:
                torch.mm(x2d, w_ih.t(), out=gi)
            else:
                torch.addmm(bias, x2d, w_ih.t(), out=gi)
            with _timed("lstm_seq", rows=n, steps=steps):
                check(lib.ctgcn_lstm_seq_f32(n, steps, hid, ptr(gi), ptr(w_hh), ptr(ln_w), ptr(ln_b), eps, 1 if reduce_sum else 0,
                                             ptr(out[lo:lo + n]), None, _stream()), "ctgcn_lstm_seq_f32")
    return out


class _LstmSeq(torch.autograd.Function):
    """LayerNorm(sum_t LSTM(seq)_t) or LayerNorm(LSTM(seq)) with autograd (rnn_type = 'LSTM', layers.py:27-28 / models.py:234-235).
    Forward = the inference kernels; backward recomputes gates, cell states and the raw h sequence chunk by chunk
    (ctgcn_lstm_seq_f32 with gates_out), runs ctgcn_layernorm_bwd_f32 and ctgcn_lstm_seq_bwd_f32 and leaves d x / d W — plain GEMMs
    over the materialised d gi — to the library.  (MIOpen's LSTM needs ~27 000 tensor-op launches per config-5 window.)"""

    @staticmethod
    def forward(ctx, seq, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b, eps, reduce_sum):
        bias = (b_ih.detach() + b_hh.detach()) if b_ih is not None else None
        seq_c = seq.contiguous()
        out = _lstm_forward(seq_c, w_ih.detach(), w_hh.detach().contiguous(), bias, ln_w, ln_b, eps, reduce_sum)
        ctx.save_for_backward(seq_c, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b)
        ctx.eps, ctx.reduce_sum = eps, reduce_sum
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        seq, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b = ctx.saved_tensors
        reduce_sum, eps = ctx.reduce_sum, ctx.eps
        rows, steps, d_in = seq.shape
        hid = w_hh.shape[1]
        dev = seq.device
        if rows == 0:
            return _zero_rnn_grads(seq, w_ih, w_hh, b_ih, b_hh, ln_w, ln_b)
        w_ih_d, w_hh_d = w_ih.detach(), w_hh.detach().contiguous()
        bias = (b_ih.detach() + b_hh.detach()) if b_ih is not None else None
        dout = dout.contiguous()
        dseq = torch.empty_like(seq)
        dw_ih, dw_hh = torch.zeros_like(w_ih_d), torch.zeros_like(w_hh_d)
        db = torch.zeros(4 * hid, dtype=torch.float32, device=dev)
        bias_part = torch.empty(512, 4 * hid, dtype=torch.float32, device=dev)
        dln_w = torch.zeros(hid, dtype=torch.float32, device=dev) if ln_w is not None else None
        dln_b = torch.zeros(hid, dtype=torch.float32, device=dev) if ln_w is not None else None
        ln_part = torch.empty(2048, 2 * hid, dtype=torch.float32, device=dev) if ln_w is not None else None
        chunks = _row_chunks(lib, rows, steps, hid)
        cmax = chunks[0][1]
        gi_buf = torch.empty(cmax * steps, 4 * hid, dtype=torch.float32, device=dev)          # reused as d_gi
        gates_buf = torch.empty(cmax * steps, 5 * hid, dtype=torch.float32, device=dev)
        hseq_buf = torch.empty(cmax, steps, hid, dtype=torch.float32, device=dev)
        hprev_buf = torch.zeros(cmax, steps, hid, dtype=torch.float32, device=dev)            # [:, 0] stays 0
        with torch.cuda.device(dev):
            for lo, n in chunks:
                x2d = seq[lo:lo + n].reshape(n * steps, d_in)
                gi, gates, hseq = gi_buf[: n * steps], gates_buf[: n * steps], hseq_buf[:n]
                if bias is None:
                    torch.mm(x2d, w_ih_d.t(), out=gi)
                else:
                    torch.addmm(bias, x2d, w_ih_d.t(), out=gi)
                check(lib.ctgcn_lstm_seq_f32(n, steps, hid, ptr(gi), ptr(w_hh_d), None, None, 0.0, 0, ptr(hseq), ptr(gates), _stream()),
                      "ctgcn_lstm_seq_f32")
                g_out = dout[lo:lo + n]
                if ln_w is not None:
                    ln_rows = n if reduce_sum else n * steps
                    dpre = torch.empty((n, hid) if reduce_sum else (n, steps, hid), dtype=torch.float32, device=dev)
                    check(lib.ctgcn_layernorm_bwd_f32(ln_rows, steps if reduce_sum else 1, hid, ptr(hseq), ptr(g_out), 0, ptr(ln_w.detach()), eps,
                                                      ptr(dpre), ptr(ln_part), ln_part.shape[0], None, _stream()), "ctgcn_layernorm_bwd_f32")
                    ln_sum = ln_part.sum(0)
                    dln_w += ln_sum[:hid]
                    dln_b += ln_sum[hid:]
                else:
                    dpre = g_out
                dgi = gi                                                                        # gi is dead: reuse as d_gi
                check(lib.ctgcn_lstm_seq_bwd_f32(n, steps, hid, ptr(gates), None if reduce_sum else ptr(dpre), ptr(dpre) if reduce_sum else None,
                                                 ptr(w_hh_d), ptr(dgi), ptr(bias_part), bias_part.shape[0], _stream()), "ctgcn_lstm_seq_bwd_f32")
                torch.mm(dgi, w_ih_d, out=dseq[lo:lo + n].view(n * steps, d_in))
                db += bias_part.sum(0)
                _accumulate_tn(dw_ih, dgi, x2d)
                hprev = hprev_buf[:n]
                hprev[:, 1:] = hseq[:, :-1]
                _accumulate_tn(dw_hh, dgi, hprev.view(n * steps, hid))
        db_ih = db_hh = None
        if b_ih is not None:
            db_ih, db_hh = db, db.clone()
        return dseq, dw_ih, dw_hh, db_ih, db_hh, dln_w, (dln_b if ln_b is not None else None), None, None


def lstm_sequence(rnn, seq, norm, reduce_sum):
    """LayerNorm(sum_t LSTM(seq)_t) / LayerNorm(LSTM(seq)) with the recurrence in ctgcn_lstm_seq_f32; with autograd enabled the
    backward runs ctgcn_lstm_seq_bwd_f32 (see _LstmSeq)."""
    b_ih = rnn.bias_ih_l0 if rnn.bias else None
    b_hh = rnn.bias_hh_l0 if rnn.bias else None
    ln_w = None if norm is None else norm.weight
    ln_b = None if norm is None else norm.bias
    eps = 0.0 if norm is None else float(norm.eps)
    return _LstmSeq.apply(seq, rnn.weight_ih_l0, rnn.weight_hh_l0, b_ih, b_hh, ln_w, ln_b, eps, bool(reduce_sum))


def gru_sequence(rnn, seq, norm, reduce_sum, out=None):
    """LayerNorm(sum_t GRU(seq)_t) (reduce_sum) or LayerNorm(GRU(seq)) — layers.py:59-62 / models.py:249-250.
    seq [rows, steps, d_in].  Input projection (ctgcn_gru_input_proj_f32), then the recurrence, the sum over steps and the
    LayerNorm in ONE HIP kernel (ctgcn_gru_seq_f32); with autograd enabled the backward runs ctgcn_gru_seq_bwd_f32 (see
    _GruSeq).  out (inference, reduce_sum): a [rows, hidden] view with unit column stride that receives the result,
    e.g. column t of the temporal GRU's [rows, T, hidden] input."""
    b_ih = rnn.bias_ih_l0 if rnn.bias else None
    b_hh = rnn.bias_hh_l0 if rnn.bias else None
    ln_w = None if norm is None else norm.weight
    ln_b = None if norm is None else norm.bias
    eps = 0.0 if norm is None else float(norm.eps)
    if out is not None:
        if torch.is_grad_enabled() and (seq.requires_grad or any(p.requires_grad for p in rnn.parameters())):
            raise RuntimeError("gru_sequence(out=...) is an inference path")
        bias, b_hn = _gru_bias(rnn, rnn.hidden_size)
        return _gru_forward(seq.contiguous(), rnn.weight_ih_l0.detach(), rnn.weight_hh_l0.detach().contiguous(), bias, b_hn,
                            ln_w, ln_b, eps, bool(reduce_sum), out=out)
    return _GruSeq.apply(seq, rnn.weight_ih_l0, rnn.weight_hh_l0, b_ih, b_hh, ln_w, ln_b, eps, bool(reduce_sum))
