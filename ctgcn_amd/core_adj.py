"""CoreAdj — one snapshot's list of k-core adjacency matrices in the fused, slot-tagged CSR form
the HIP aggregation kernel streams (include/ctgcn_hip.h: ctgcn_core_aggregate_f32).

The reference keeps K separate N x N torch sparse COO tensors per snapshot (helper.py:51-82, 20 B per
entry each).  Because k-core subgraphs are nested (A_kmax ⊆ ... ⊆ A_1) the same information is ONE CSR
over the largest kept matrix plus a uint8 tag per entry: entry tagged s is present in matrices
s, s+1, ..., K-1 ("slot" = position in the reference's adjacency list).  9 B per entry of the largest
matrix instead of 20 B per entry of every matrix.  Lists that are not nested (arbitrary user supplied
matrices) are kept too: then an entry is stored once per matrix containing it and the NESTED flag is off.

Two builders:
  * CoreAdj.from_matrices(...)      host side, from scipy matrices / torch sparse tensors (the .npz route)
  * CoreAdj.from_graph(...)         device side, from the snapshot's CSR: k-core peel -> edge levels ->
                                    slot table (host, O(max_core)) -> per-row stable partition, all HIP.
"""
import numpy as np
import torch

from . import _lib

_SENTINEL = 255


def _as_i32_words(words):
    """int64 tensor of 32-bit patterns -> int32 with the same bits (bit 31 set: negative)"""
    w = words & 0xffffffff
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()


class CoreAdj(object):
    """Slot-tagged CSR of the K matrices the reference loader would return for one snapshot.

    Quacks like the reference's inner list where it matters: len() == K, iteration yields torch sparse
    COO tensors equal to the reference's (built lazily, for interop only — the models never use them).
    """

    LONG_ROW = 2048     # rows with more stored entries than this are "hubs": one block per row instead of one lane group

    def __init__(self, n, K, row_ptr, col, val, slot, self_loop, nested, symmetric, nnz_per_slot, levels=None):
        self.n, self.K = int(n), int(K)
        self.row_ptr, self.col, self.val, self.slot = row_ptr, col, val, slot
        self.self_loop, self.nested, self.symmetric = bool(self_loop), bool(nested), bool(symmetric)
        self.nnz_per_slot = [int(v) for v in nnz_per_slot]      # reference-semantics nnz(A_j), incl. the +I of slot 0
        self.levels = None if levels is None else [int(v) for v in levels]   # k value of each slot (k-core route)
        self._t = None                                           # transposed arrays, built on demand when not symmetric
        self._moved = {}                                         # copies on other devices (as_core_adj: moved once, not per call)
        self._long = {}                                          # cached hub-row lists (forward / transposed)
        self._plan = None                                        # cached row plan of the inference path (row_plan)
        assert 0 <= self.K <= _lib.MAX_SLOTS

    # ------------------------------------------------------------------ list-like surface
    def __len__(self):
        return self.K

    def __iter__(self):
        return iter(self.to_torch_sparse_list())

    @property
    def nnz(self):
        return int(self.col.numel())

    @property
    def device(self):
        return self.col.device

    @property
    def flags(self):
        return (_lib.F_SELF_LOOP if self.self_loop else 0) | (_lib.F_NESTED if self.nested else 0)

    @property
    def aggregated_edges(self):
        """sum_k nnz(A_k): the numerator of the 'aggregated edges/s' metric for one CoreDiffusion call."""
        return sum(self.nnz_per_slot)

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        if device == self.device:
            return self
        hit = self._moved.get(str(device))
        if hit is not None:
            return hit
        out = self._moved[str(device)] = self._copy_to(device)
        return out

    def _copy_to(self, device):
        out = CoreAdj(self.n, self.K, self.row_ptr.to(device), self.col.to(device), self.val.to(device),
                      self.slot.to(device), self.self_loop, self.nested, self.symmetric, self.nnz_per_slot, self.levels)
        if self._t is not None:
            out._t = tuple(a.to(device) for a in self._t)
        return out

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def cpu(self):
        return self.to("cpu")

    HUB_SPLIT_MAX = 32

    @property
    def HUB_SPLIT_ENTRIES(self):
        """entries per block when a hub row is cut into pieces: the library's constant (ctgcn_hub_split_entries), not a copy of it"""
        return int(_lib.load().ctgcn_hub_split_entries())

    def hub_split(self, transposed=False):
        """Blocks per hub row the aggregation kernels may use (1: no row is long enough to be cut into pieces)."""
        key = ("split", bool(transposed) and not self.symmetric)
        if key not in self._long:
            lr = self.long_rows(transposed)
            if lr is None:
                self._long[key] = 1
            else:
                rp = self.transposed()[0] if key[1] else self.row_ptr
                longest = int((rp[1:] - rp[:-1]).max().item())
                self._long[key] = max(1, min(self.HUB_SPLIT_MAX, -(-longest // self.HUB_SPLIT_ENTRIES)))
        return self._long[key]

    def long_rows(self, transposed=False):
        """int32 device tensor of the rows with more than LONG_ROW entries (None if there are none)."""
        key = bool(transposed) and not self.symmetric
        if key not in self._long:
            rp = self.transposed()[0] if key else self.row_ptr
            idx = torch.nonzero((rp[1:] - rp[:-1]) > self.LONG_ROW).flatten().to(torch.int32)
            self._long[key] = idx if idx.numel() else None
        return self._long[key]

    # ------------------------------------------------------------------ repeated rows of H (inference path)
    PLAN_TILE = 16      # sequences per tile of the GRU layer kernel (gru_layer8_h2_kernel)
    PLAN_TILE_GEMM = 64 # ... and of the recurrence kernel behind the split GEMM (gru_seq_h2_kernel)

    def row_plan(self, tile=None):
        """Row plan for the inference path (ctgcn_core_aggregate_split_f32 and its consumers), or None (K > 64).

        layers.py:41-48: res_j = res_{j-1} + A_j x, so as long as no entry of row v has arrived (slots below the row's first tag f;
        nested lists: f = K - capped core number) H[v, 0..f-1] is the same row relu(x_v) f times, and the GRU multiplies it by W_ih f
        times (layers.py:58-59).  The plan lets the kernels write and multiply it once:
          order      int32[n]   matrix row handled at position p.  Rows with equal repeat patterns are neighbours, so that the
                                sequences of a GRU tile share one; inside a pattern rows go by falling degree (the two rows of a
                                wave and the eight of a block are equally long, and the long blocks of a launch start first).
          tile_mask  int32[ceil(n / tile)]  bit j set = slot j carries a new row for at least one of the tile's positions (bit 0 always);
                                K > 32 (America-Air: 64, Europe-Air: 33): int32[2 tiles], word 2 T = slots 0-31, word 2 T + 1 = slots 32-63
          tile_base  int32[tiles]  first COMPACT operand row of the tile (tile * popcount(mask) rows per tile, incl. the padding
                                of the last one) and operand_rows, their total: the GEMM consumer's layout (tile = 64)
          inverse    int32[n]   position of matrix row v (hub rows are looked up here)
        tile: PLAN_TILE (16: the GRU layer kernel reads the planes with holes) or PLAN_TILE_GEMM (64).
        Static per graph: built once on the device, cached."""
        tile = self.PLAN_TILE if tile is None else int(tile)
        if self.K > 64 or self.n == 0:
            return None
        if self._plan is None:
            self._plan = {}
        if "order" not in self._plan:
            dev, n, K = self.device, self.n, self.K
            rp = self.row_ptr.long()
            deg = rp[1:] - rp[:-1]
            if self.nested:
                # entries are sorted by (row, slot): the first entry of a row carries its smallest tag; every later slot adds P != 0
                first = torch.full((n,), K, dtype=torch.int64, device=dev)
                has = deg > 0
                first[has] = self.slot[rp[:-1][has]].long()
                # bits first .. K-1 (first = K: none) | bit 0 — arithmetic on n values, no [n, K] temporaries.  Two 32-bit halves (in int64):
                # half w holds slots 32 w .. 32 w + 31
                def half(w):
                    lo, hi = 32 * w, min(K, 32 * w + 32)
                    if hi <= lo:
                        return torch.zeros_like(first)
                    f = (first - lo).clamp(0, hi - lo)
                    return (1 << (hi - lo)) - (torch.ones_like(first) << f)
                mask_lo, mask_hi = half(0) | 1, half(1)
            else:
                rows = torch.repeat_interleave(torch.arange(n, device=dev), deg)
                mask_lo = torch.ones(n, dtype=torch.int64, device=dev)
                mask_hi = torch.zeros(n, dtype=torch.int64, device=dev)
                for j in range(K):                                        # one [n] bool per slot instead of an [n, K] product
                    hit = torch.zeros(n, dtype=torch.bool, device=dev)
                    hit[rows[self.slot == j]] = True
                    if j < 32:
                        mask_lo |= hit.long() << j
                    else:
                        mask_hi |= hit.long() << (j - 32)
            o1 = torch.argsort(deg, descending=True, stable=True)
            # rows with equal patterns become neighbours: by the 64-bit pattern (hi, lo), falling — two stable sorts (K <= 32: hi = 0, the one sort of before)
            order = o1[torch.argsort(mask_lo[o1], descending=True, stable=True)]
            if K > 32:
                order = order[torch.argsort(mask_hi[order], descending=True, stable=True)]
            inverse = torch.empty(n, dtype=torch.int64, device=dev)
            inverse[order] = torch.arange(n, device=dev)
            self._plan.update(order=order.to(torch.int32).contiguous(), inverse=inverse.to(torch.int32).contiguous(),
                              _sorted_mask=mask_lo[order], _sorted_mask_hi=mask_hi[order])
        if tile not in self._plan:
            dev, n, K = self.device, self.n, self.K
            ntiles = -(-n // tile)
            padded = torch.ones(ntiles * tile, dtype=torch.int64, device=dev)
            padded[:n] = self._plan["_sorted_mask"]
            tiles = padded.view(ntiles, tile)
            padded_hi = torch.zeros(ntiles * tile, dtype=torch.int64, device=dev)
            padded_hi[:n] = self._plan["_sorted_mask_hi"]
            tiles_hi = padded_hi.view(ntiles, tile)
            tmask = torch.zeros(ntiles, dtype=torch.int64, device=dev)
            tmask_hi = torch.zeros(ntiles, dtype=torch.int64, device=dev)
            fresh = torch.zeros(ntiles, dtype=torch.int64, device=dev)
            for j in range(K):
                anyj = (((tiles if j < 32 else tiles_hi) >> (j & 31)) & 1).any(1).long()
                if j < 32:
                    tmask |= anyj << j
                else:
                    tmask_hi |= anyj << (j - 32)
                fresh += anyj
            if K > 32:                                                # two words per tile, interleaved
                tmask = torch.stack([tmask, tmask_hi], 1).reshape(-1)
            per_tile = fresh * tile                                   # compact operand rows of each tile
            base = torch.cumsum(per_tile, 0) - per_tile
            total = int(per_tile.sum().item())
            if total >= 2 ** 31:
                return None
            self._plan[tile] = dict(order=self._plan["order"], inverse=self._plan["inverse"], tile=tile,
                                    tile_mask=_as_i32_words(tmask), tile_base=base.to(torch.int32).contiguous(),
                                    new_rows=total, operand_rows=total)      # (position, slot) rows written per layer (incl. tile padding)
        return self._plan[tile]

    def plan_row_dest(self, plan, rows, compact):
        """int32[len(rows) * K]: operand row of slot j of matrix row rows[i] under `plan` (hub rows take this map), -1 = not wanted.
        compact: the GEMM consumer's layout (tile_base), else rows position * K + slot."""
        K, tile = self.K, plan["tile"]
        key = ("dest", tile, bool(compact), rows.data_ptr(), int(rows.numel()))       # static per (graph, tile, layout): built once
        hit = self._plan.get(key) if self._plan is not None else None
        if hit is not None:
            return hit
        pos = plan["inverse"][rows.long()].long()
        t = torch.div(pos, tile, rounding_mode="floor")
        j = torch.arange(K, device=pos.device)[None, :]
        if K > 32:
            lo, hi = plan["tile_mask"][2 * t].long() & 0xffffffff, plan["tile_mask"][2 * t + 1].long() & 0xffffffff
            bits = torch.where(j < 32, lo[:, None] >> j.clamp(max=31), hi[:, None] >> (j - 32).clamp(min=0)) & 1
        else:
            need = plan["tile_mask"][t].long() & 0xffffffff
            bits = (need[:, None] >> j) & 1
        if compact:
            rank = torch.cumsum(bits, 1) - bits
            dest = plan["tile_base"][t].long()[:, None] + (pos % tile)[:, None] * bits.sum(1, keepdim=True) + rank
        else:
            dest = pos[:, None] * K + j
        out = torch.where(bits.bool(), dest, torch.full_like(dest, -1)).to(torch.int32).contiguous().view(-1)
        if self._plan is not None:
            self._plan[key] = out
        return out

    # ------------------------------------------------------------------ transposed view (backward pass)
    def transposed(self):
        """(row_ptr, col, val, slot) of the transposed matrices, same (row, slot, col) ordering."""
        if self.symmetric:
            return self.row_ptr, self.col, self.val, self.slot
        if self._t is None:
            dev = self.device
            rows = torch.repeat_interleave(torch.arange(self.n, device=dev), (self.row_ptr[1:] - self.row_ptr[:-1]).long())
            key = (self.col.long() * (self.K + 1) + self.slot.long()) * self.n + rows
            order = torch.argsort(key, stable=True)
            t_rows = self.col.long()[order]
            counts = torch.bincount(t_rows, minlength=self.n)
            t_ptr = torch.zeros(self.n + 1, dtype=torch.int64, device=dev)
            t_ptr[1:] = torch.cumsum(counts, 0)
            self._t = (t_ptr.to(torch.int32), rows[order].to(torch.int32).contiguous(), self.val[order].contiguous(),
                       self.slot[order].contiguous())
        return self._t

    # ------------------------------------------------------------------ interop / tests
    def to_scipy_list(self):
        """The K matrices as scipy CSR float32, exactly what the reference loader's list holds (coalesced)."""
        import scipy.sparse as sp
        rp = self.row_ptr.cpu().numpy().astype(np.int64)
        col = self.col.cpu().numpy()
        val = self.val.cpu().numpy()
        slot = self.slot.cpu().numpy().astype(np.int64)
        rows = np.repeat(np.arange(self.n), np.diff(rp))
        out = []
        for j in range(self.K):
            keep = (slot <= j) if self.nested else (slot == j)
            m = sp.csr_matrix((val[keep], (rows[keep], col[keep])), shape=(self.n, self.n), dtype=np.float32)
            if j == 0 and self.self_loop:
                m = (m + sp.eye(self.n, dtype=np.float32, format="csr")).tocsr()
            m.sum_duplicates()
            m.sort_indices()
            out.append(m)
        return out

    def to_torch_sparse_list(self):
        out = []
        for m in self.to_scipy_list():
            coo = m.tocoo()
            idx = torch.from_numpy(np.vstack((coo.row, coo.col)).astype(np.int64))
            out.append(torch.sparse_coo_tensor(idx, torch.from_numpy(coo.data), coo.shape).to(self.device))
        return out

    # ------------------------------------------------------------------ builder 1: host, from explicit matrices
    @staticmethod
    def from_matrices(mats, self_loop=None, device="cpu"):
        """mats: the K matrices in list order (scipy sparse, or torch sparse COO/CSR tensors).

        If `self_loop` is None, a first matrix of the form B + I (unit diagonal, B and every later matrix
        zero-diagonal — what helper.py:71-72 produces) is detected and the diagonal is folded into the
        SELF_LOOP flag.  Nestedness (identical entries forming a suffix of the list) is detected; lists
        that are not nested are stored entry-per-matrix with NESTED off.
        """
        import scipy.sparse as sp
        K = len(mats)
        if K == 0:
            raise ValueError("empty adjacency list")
        if K > _lib.MAX_SLOTS:
            raise ValueError("at most %d matrices per snapshot" % _lib.MAX_SLOTS)
        sm = []
        for m in mats:
            if isinstance(m, torch.Tensor):
                m = m.detach().cpu()
                if m.layout != torch.sparse_coo:
                    m = m.to_sparse_coo() if m.layout != torch.strided else m.to_sparse()
                m = m.coalesce()
                ii = m.indices().numpy()
                m = sp.csr_matrix((m.values().numpy().astype(np.float32), (ii[0], ii[1])), shape=tuple(m.shape))
            else:
                m = sp.csr_matrix(m).astype(np.float32)
                m.sum_duplicates()
            sm.append(m)
        n = sm[0].shape[0]
        for m in sm:
            if m.shape != (n, n):
                raise ValueError("adjacency matrices must all be %d x %d" % (n, n))
        nnz_per_slot = [int(m.nnz) for m in sm]

        if self_loop is None:
            d0 = sm[0].diagonal()
            self_loop = bool(n > 0 and np.all(d0 == 1.0) and not any(m.diagonal().any() for m in sm[1:]))
            strip = self_loop
        else:
            strip = False   # caller passes B and asks for B + I
            if self_loop:
                nnz_per_slot[0] += n
        if strip:
            first = sm[0].tolil()
            first.setdiag(0)
            first = first.tocsr()
            first.eliminate_zeros()
            sm[0] = first

        rows = np.concatenate([m.tocoo().row for m in sm]).astype(np.int64)
        cols = np.concatenate([m.tocoo().col for m in sm]).astype(np.int64)
        vals = np.concatenate([m.tocoo().data for m in sm]).astype(np.float32)
        tags = np.concatenate([np.full(m.nnz, j, dtype=np.int64) for j, m in enumerate(sm)])

        # group identical (row, col); nested  <=>  every group is {s, s+1, ..., K-1} with one value
        order = np.lexsort((tags, cols, rows))
        rows, cols, vals, tags = rows[order], cols[order], vals[order], tags[order]
        key = rows * n + cols
        first_of_group = np.ones(len(key), dtype=bool)
        first_of_group[1:] = key[1:] != key[:-1]
        gid = np.cumsum(first_of_group) - 1
        nested = True
        if len(key):
            gstart = np.flatnonzero(first_of_group)
            gsize = np.diff(np.append(gstart, len(key)))
            smin = tags[gstart]
            if not np.array_equal(gsize, K - smin):      # a suffix has exactly K - s members (tags are distinct & sorted)
                nested = False
            elif not np.array_equal(vals, vals[gstart][gid]):
                nested = False
        if nested:
            keep = first_of_group
            rows, cols, vals, tags = rows[keep], cols[keep], vals[keep], tags[keep]
        order = np.lexsort((cols, tags, rows))
        rows, cols, vals, tags = rows[order], cols[order], vals[order], tags[order]
        row_ptr = np.zeros(n + 1, dtype=np.int64)
        np.cumsum(np.bincount(rows, minlength=n), out=row_ptr[1:])
        if row_ptr[-1] >= 2 ** 31:
            raise ValueError("more than 2^31-1 stored entries")

        # symmetric (structure, value AND tag) -> the backward pass can reuse the same arrays
        symmetric = False
        if nested:
            mv = sp.csr_matrix((vals, (rows, cols)), shape=(n, n))
            mt = sp.csr_matrix((tags + 1, (rows, cols)), shape=(n, n))
            symmetric = bool((mv != mv.T).nnz == 0 and (mt != mt.T).nnz == 0)
        adj = CoreAdj(n, K, torch.from_numpy(row_ptr.astype(np.int32)), torch.from_numpy(cols.astype(np.int32)),
                      torch.from_numpy(vals), torch.from_numpy(tags.astype(np.uint8)), self_loop, nested, symmetric,
                      nnz_per_slot)
        return adj.to(device)

    # ------------------------------------------------------------------ builder 1b: device, from NESTED k-core matrices
    @staticmethod
    def from_nested_matrices_device(mats, device, self_loop=True):
        """The .npz loader route at scale: `mats` = the kept k-core matrices of one snapshot in list order (scipy sparse,
        smallest first: A_kmax ⊆ ... ⊆ A_1, zero diagonal, as helper.py:63-78 visits them; slot 0 gets + I through the
        SELF_LOOP flag).  The K coordinate lists are uploaded once and tagged on the GPU: an entry's slot is K minus the number
        of matrices that contain it (one sort + run-length count), then one more sort puts rows in (slot, col) order — instead
        of from_matrices' host lexsort over all K·nnz entries.  Returns None when the list turns out not to be nested (or the
        values differ between matrices): the caller then uses the general host builder."""
        import scipy.sparse as sp
        K = len(mats)
        if K == 0 or K > _lib.MAX_SLOTS:
            return None
        n = mats[0].shape[0]
        dev = torch.device(device)
        keys, vals, nnz_per_slot = [], [], []
        for m in mats:
            m = sp.coo_matrix(m)
            if m.shape != (n, n):
                raise ValueError("adjacency matrices must all be %d x %d" % (n, n))
            keys.append(torch.from_numpy(m.row.astype(np.int64) * n + m.col.astype(np.int64)))
            vals.append(torch.from_numpy(m.data.astype(np.float32)))
            nnz_per_slot.append(int(m.nnz))
        if self_loop:
            nnz_per_slot[0] += n
        key_all = torch.cat(keys).to(dev)
        val_all = torch.cat(vals).to(dev)
        uniq, inverse, counts = torch.unique(key_all, sorted=True, return_inverse=True, return_counts=True)
        if uniq.numel() != mats[-1].nnz or uniq.numel() >= 2 ** 31:
            return None                                   # an entry outside the largest matrix (or duplicates inside one): not nested
        # one value per entry: every copy must equal the copy in the largest matrix
        last_lo = key_all.numel() - mats[-1].nnz
        val_of = torch.empty(uniq.numel(), dtype=torch.float32, device=dev)
        val_of[inverse[last_lo:]] = val_all[last_lo:]
        if not bool((val_of[inverse] == val_all).all()):
            return None
        rows = torch.div(uniq, n, rounding_mode="floor")
        cols = uniq - rows * n
        if bool((rows == cols).any()):
            return None                                   # a stored diagonal: the + I fold below would be wrong
        slot = (K - counts).to(torch.int64)               # nested: present in matrices slot .. K-1
        # nestedness proper: the copies of an entry must sit in the LAST `count` matrices
        mat_of = torch.repeat_interleave(torch.arange(K, device=dev), torch.tensor([k.numel() for k in keys], device=dev))
        if not bool((mat_of >= slot[inverse]).all()):
            return None
        order = torch.argsort((rows * K + slot) * n + cols)
        rows, cols, slot, val_of = rows[order], cols[order], slot[order], val_of[order]
        row_ptr = torch.zeros(n + 1, dtype=torch.int64, device=dev)
        row_ptr[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
        # symmetric in structure, value and tag?  (always for the reference's k-core files)  key of the transposed entry
        tkey = cols * n + rows
        pos = torch.searchsorted(uniq, tkey)
        pos_ok = pos < uniq.numel()
        symmetric = bool(pos_ok.all()) and bool((uniq[pos.clamp(max=uniq.numel() - 1)] == tkey).all())
        if symmetric:
            inv_order = torch.empty_like(order)
            inv_order[order] = torch.arange(order.numel(), device=dev)
            mate = inv_order[pos]                         # position (in the final order) of the transposed entry
            symmetric = bool((val_of[mate] == val_of).all()) and bool((slot[mate] == slot).all())
        return CoreAdj(n, K, row_ptr.to(torch.int32), cols.to(torch.int32).contiguous(), val_of.contiguous(),
                       slot.to(torch.uint8).contiguous(), bool(self_loop), True, symmetric, nnz_per_slot)

    # ------------------------------------------------------------------ builder 2: device, from the snapshot graph
    @staticmethod
    def from_graph(row_ptr, col, val, max_core=-1, core=None):
        """Device route.  row_ptr/col/val: CUDA tensors holding the snapshot's symmetric, zero-diagonal CSR
        (int32/int32/float32).  Runs the HIP k-core peel (unless `core` is given), tags every entry with
        level = min(core[u], core[v]) and applies the loader semantics of helper.py:51-82 through a
        level -> slot table.  Returns (CoreAdj, core_numbers, file_count) where file_count is the number
        of per-k files the reference would have written (= max core number), which the caller needs for
        the sticky max_core rule (helper.py:61-62).  With max_core = L >= 1 the loader never tells levels above L
        apart (helper.py:63), so the peel stops at L: core numbers and file_count are then reported capped at L.
        """
        from . import ops
        n = row_ptr.numel() - 1
        if max_core == 0:
            # an edgeless first snapshot under max_core=-1 makes the sticky value 0: the reference then keeps f_list[:0] = []
            # for every later snapshot (helper.py:61-64)
            if core is None:
                core, max_k = ops.kcore(row_ptr, col)
            else:
                max_k = int(core.max().item()) if n else 0
            return None, core, max_k
        if core is None:
            core, max_k = ops.kcore(row_ptr, col, level_cap=max_core if max_core >= 1 else -1)
        else:
            max_k = int(core.max().item()) if n else 0
        file_count = max_k
        if file_count == 0:
            return None, core, 0
        hist_len = file_count + 1
        level, count, wsum = ops.edge_levels(row_ptr, col, val, core, hist_len)
        table, K, levels, nnz_per_slot = slot_table(count.cpu().numpy(), wsum.cpu().numpy(), file_count, max_core, n)
        col2, val2, slot2 = ops.slot_reorder(row_ptr, col, val, level, torch.from_numpy(table).to(col.device), K)
        adj = CoreAdj(n, K, row_ptr, col2, val2, slot2, True, True, True, nnz_per_slot, levels)
        return adj, core, file_count


def slot_table(count, wsum, file_count, max_core, n):
    """Loader semantics of helper.py:58-78 as a level -> slot table.

    count[L] / wsum[L]: number / weight-sum of CSR entries whose level min(core[u],core[v]) is L
    (L = 0..file_count).  Files 1..file_count exist (structure_generation.py:47-56); the loader keeps
    k = 1..min(max_core, file_count) (helper.py:63) and visits them from the largest k down (helper.py:64).
    Slot 0 = A_kept + I.  Going down, matrix A_k is dropped when (A_k - A_{k+1}).sum() == 0, i.e. when the
    entries of level exactly k sum to zero (helper.py:74-76).  Returns (table uint8[file_count+1], K,
    k-value per slot, reference nnz per slot).
    """
    count = np.asarray(count, dtype=np.int64)
    wsum = np.asarray(wsum, dtype=np.float64)
    kept = file_count if max_core < 0 else min(int(max_core), file_count)
    if kept < 1:
        raise ValueError("max_core must keep at least one k-core matrix")
    table = np.full(file_count + 1, _SENTINEL, dtype=np.uint8)
    table[kept:] = 0
    levels = [kept]
    nnz = [int(count[kept:].sum()) + n]
    for k in range(kept - 1, 0, -1):
        if wsum[k] == 0:
            if count[k] != 0:
                raise NotImplementedError(
                    "entries of k-core level %d have weights that sum to zero; the reference drops that matrix "
                    "but keeps its entries for later ones — use the .npz route (CoreAdj.from_matrices)" % k)
            continue
        if len(levels) >= _lib.MAX_SLOTS:
            raise ValueError("more than %d distinct k-core matrices in one snapshot" % _lib.MAX_SLOTS)
        table[k] = len(levels)
        levels.append(k)
        nnz.append(int(count[k:].sum()))
    assert count[0] == 0, "an edge endpoint cannot have core number 0"
    return table, len(levels), levels, nnz
