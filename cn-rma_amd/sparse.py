"""Sparse-voxel tensors and operators on the HIP engine (csrc/sparse.hip) -- the replacement of the
MinkowskiEngine v0.5.4 surface the reference uses:

    ME.utils.batch_sparse_collate + ME.SparseTensor      -> voxelize()
    MinkowskiConvolution (k3/k1, stride 1/2)              -> conv()
    MinkowskiGenerativeConvolutionTranspose (k2 s2)       -> conv_transpose_generative()
    MinkowskiMaxPooling (k2 s2)                           -> max_pool()
    MinkowskiInstanceNorm / BatchNorm / ReLU / ELU        -> instance_norm() / fused conv epilogues
    A + B on different coordinate sets                    -> union_add()
    SparseTensor.features_at_coordinates                  -> interpolate()
    MinkowskiPruning                                      -> prune()

Reference call sites: projects/mvsdetection/models/ray_marching.py:328-330, fcaf3d_backbone.py:26-31,63-70,
fcaf3d_head.py:64-98,107-139,275-298.

A CoordSet owns the int32 coordinates [N,4] = (batch,x,y,z) of one lattice level, its hash map and the cached
neighbour tables / strided children (what ME's CoordinateManager caches); a SparseTensor is (CoordSet, features).
"""
import itertools

import os
import threading

import torch
from torch.utils.weak import WeakIdKeyDictionary

from . import _lib
from . import plan as P
from ._lib import call, ptr, stream


def _next_pow2(n):
    p = 2
    while p < n:
        p *= 2
    return p


def kernel_offsets(kernel_size, tensor_stride):
    """ME kernel-offset order: linear index decodes with x fastest; odd k centred, even k = 0..k-1 (Appendix A)."""
    k = kernel_size
    rng = [(i - k // 2) if k % 2 == 1 else i for i in range(k)]
    offs = [(ix * tensor_stride, iy * tensor_stride, iz * tensor_stride)
            for iz, iy, ix in itertools.product(rng, rng, rng)]
    return offs


class CoordMap:
    """Open-addressing hash table on the device: uint64 key -> int32 row."""

    def __init__(self, n, device, arena=False):
        """arena: take the keys from the static trace's pre-cleared 0xFF arena when there is one (self.precleared)"""
        self.cap = _next_pow2(max(2 * n, 16))
        plan = P.current() if arena else None
        a = plan.arena_ff(self.cap * 8, device) if plan is not None else None
        self.precleared = a is not None
        self.keys = a[:self.cap * 8].view(torch.int64) if a is not None else \
            torch.empty(self.cap, dtype=torch.int64, device=device)            # reinterpreted as uint64
        self.vals = torch.empty(self.cap, dtype=torch.int32, device=device)


_OFFSETS = {}


def _offsets_tensor(kernel_size, stride, device):
    key = (kernel_size, stride, device)
    if key not in _OFFSETS:
        _OFFSETS[key] = torch.tensor(kernel_offsets(kernel_size, stride), dtype=torch.int32, device=device)
    return _OFFSETS[key]


def scene_counts(scene_col, n_scenes):
    """rows per scene id as host ints with ONE device->host read (torch.bincount reads the maximum back first)"""
    return _lib.read_ints(torch.stack([(scene_col == b).sum() for b in range(n_scenes)]))


def _count(n_out, bound, lo=0):
    """row count of a tensor a kernel has just produced (n_out: device int32 [1]) -> (n, n_dev).
    Eager: the count is read back (n exact, n_dev None) and recorded when a Plan is calibrating; static trace: n is the
    planned CAPACITY and the live count stays on the device (n_dev = n_out), registered as `lo <= n_out <= capacity`."""
    plan = P.current()
    if plan is not None and plan.static:
        return plan.next_cap(bound, n_out, lo), n_out
    n = _lib.read_ints(n_out)[0]
    if n < 0:
        raise _lib.CnrmaError("voxelize: a point lies outside the coordinate-key range (|coord / voxel_size| >= 32767, NaN, "
                              "or batch id >= 65536)")
    if plan is not None:
        plan.record(n)
    return n, None


class CoordSet:
    """n = number of rows (eager) or their capacity (static trace; then n_dev is the device word holding the live count
    and rows >= n_dev[0] are undefined -- every kernel gets both)."""

    def __init__(self, coords, stride, cmap=None, n_batch=1, n=None, n_dev=None):
        assert coords.dtype == torch.int32 and coords.dim() == 2 and coords.shape[1] == 4
        self.n = coords.shape[0] if n is None else int(n)
        self.n_dev = n_dev
        self.C = coords[:self.n].contiguous()
        self.stride = int(stride)
        self.device = coords.device
        self.n_batch = n_batch     # number of scenes in the tensor (the reference is structurally 1 per GPU)
        self.scene_major = n_batch <= 1   # rows of one scene contiguous and scenes in order (set by collate / strided)
        self.compact = False      # consecutive rows are spatial neighbours (Morton runs): the voxeliser's sets and what is
                                  # derived from them; selects the gather-once convolution
        self.sorted = False       # the rows are ONE run in ascending Morton key (voxeliser, strided sets, generated children,
                                  # pruned subsets): strided coordinates then need no hash table
        self._counts = None
        self._counts_dev = None
        self._map = cmap
        self._nbr = {}        # (kernel_size, id(out CoordSet)) -> nbr table
        self._union = {}      # same key -> tile unions of that table (gather-once convolution)
        self._children = {}   # new_stride -> CoordSet
        self._gen_parent = None   # set by conv_transpose_generative: the parent set whose 8 children per row these rows are
        self._offsets = {}

    def batch_counts(self):
        """rows per scene (host ints; one device->host read per coordinate set, cached)"""
        if self._counts is None:
            if self.n_batch <= 1:
                self._counts = [self.n]
            else:
                self._counts = scene_counts(self.C[:, 0], self.n_batch)
        return self._counts

    def counts_dev(self):
        """live rows per scene as a device tensor int32 [n_batch] (no read-back; cached) and their exclusive offsets
        (meaningful for scene-major row sets)"""
        if self._counts_dev is None:
            dev = self.device
            if self.n_batch <= 1:
                c = self.n_dev.view(1).to(torch.int32) if self.n_dev is not None else \
                    torch.full((1,), self.n, dtype=torch.int32, device=dev)
            else:
                # one compare against the scene ids + one column sum (a scatter_add into n_batch bins serialises on its
                # few addresses: 0.16 ms per scene at the ScanNet shape)
                sc = self.C[:, 0]
                if self.n_dev is not None:
                    live = torch.arange(self.n, device=dev, dtype=torch.int32) < self.n_dev
                    sc = torch.where(live, sc, torch.full_like(sc, -1))
                ids = torch.arange(self.n_batch, device=dev, dtype=torch.int32)
                c = (sc.view(-1, 1) == ids.view(1, -1)).sum(dim=0).to(torch.int32)
            off = (torch.cumsum(c, 0) - c).to(torch.int32)
            self._counts_dev = (c, off)
        return self._counts_dev

    @property
    def cmap(self):
        if self._map is None:
            m = CoordMap(self.n, self.device, arena=True)
            call("cnrma_sparse_build_map", ptr(self.C), self.n, ptr(self.n_dev), ptr(m.keys), ptr(m.vals), m.cap, int(m.precleared),
                 stream())
            self._map = m
        return self._map

    def strided(self, factor=2):
        """output sites of a stride-`factor` conv / pool: unique(floor(p / s') * s'), s' = factor * stride."""
        ns = self.stride * factor
        if ns not in self._children:
            self.prefetch_strided(1, factor)
        return self._children[ns]

    def prefetch_strided(self, levels, factor=2):
        """the chain self -> strided -> strided ... (`levels` deep) enqueued back to back -- level k+1 reads level k's
        row count on the device -- with ONE device->host read of all row counts at the end instead of one per level.
        The sets are cached, so the strided() calls of the layers that follow cost nothing."""
        todo, cs, ns = [], self, self.stride
        while len(todo) < levels and (ns * factor) in cs._children:        # already known prefix
            ns *= factor
            cs = cs._children[ns]
            levels -= 1
        src_C, src_ndev, cap = cs.C, cs.n_dev, cs.n
        if levels <= 0 or cap == 0:
            return
        ws = torch.empty(_lib.load().cnrma_voxelize_workspace_bytes(cap), dtype=torch.uint8, device=self.device)
        plan = P.current()
        if plan is not None and plan.static:
            # static trace: every level gets its planned capacity (hash table sized for it); the coordinate buffer keeps
            # the provable bound (rows of the level above), so a scene that outgrows the plan cannot write out of bounds
            for _ in range(levels):
                ns *= factor
                out = torch.empty((cap, 4), dtype=torch.int32, device=self.device)
                n_out = torch.empty(1, dtype=torch.int32, device=self.device)
                cap_k = plan.next_cap(cap, n_out)
                if cs.sorted:                         # adjacent comparison instead of a hash insert per input row
                    m = None
                    call("cnrma_sparse_stride_coords_sorted", ptr(src_C), cap, ptr(src_ndev), ns, ptr(out), cap_k, ptr(n_out),
                         ptr(ws), stream())
                else:
                    m = CoordMap(cap_k, self.device)
                    call("cnrma_sparse_stride_coords", ptr(src_C), cap, ptr(src_ndev), ns, ptr(m.keys), ptr(m.vals), m.cap,
                         ptr(out), cap_k, ptr(n_out), ptr(ws), stream())
                child = CoordSet(out, ns, m, self.n_batch, n=cap_k, n_dev=n_out)
                child.scene_major = cs.scene_major
                child.compact = cs.compact            # strided rows are written in the order of their keys
                child.sorted = cs.sorted
                cs._children[ns] = child
                cs, src_C, src_ndev, cap = child, child.C, n_out, cap_k
            return
        for _ in range(levels):
            ns *= factor
            out = torch.empty((cap, 4), dtype=torch.int32, device=self.device)
            n_out = torch.empty(1, dtype=torch.int32, device=self.device)
            if cs.sorted:
                m = None
                call("cnrma_sparse_stride_coords_sorted", ptr(src_C), cap, ptr(src_ndev), ns, ptr(out), 0, ptr(n_out), ptr(ws),
                     stream())
            else:
                m = CoordMap(cap, self.device)
                call("cnrma_sparse_stride_coords", ptr(src_C), cap, ptr(src_ndev), ns, ptr(m.keys), ptr(m.vals), m.cap, ptr(out),
                     0, ptr(n_out), ptr(ws), stream())
            todo.append((ns, out, n_out, m))
            src_C, src_ndev = out, n_out
        counts = _lib.read_ints(torch.cat([t[2] for t in todo]))
        if plan is not None:
            for n in counts:
                plan.record(n)
        for (ns_k, out, _, m), n in zip(todo, counts):
            if m is not None and m.cap > 8 * _next_pow2(max(2 * n, 16)):
                m = None                      # a far over-sized table scatters the probes: rebuild compactly on first use
            child = CoordSet(out[:n], ns_k, m, self.n_batch)
            child.scene_major = cs.scene_major
            child.compact = cs.compact
            child.sorted = cs.sorted
            cs._children[ns_k] = child
            cs = child

    def neighbours(self, out_set, kernel_size, offset_stride, method="auto"):
        """nbr[No][K]: row of `self` at out_coord + offset_k (or -1).  method: "generic" probes every (output, offset)
        pair; "auto" uses the symmetric builder for stride-1 odd kernels on one set and the input-driven builder for
        the stride-2 maps (identical tables, 2-7x fewer probes)."""
        key = (kernel_size, offset_stride, id(out_set))
        if key not in self._nbr:
            offs = _offsets_tensor(kernel_size, offset_stride, self.device)
            K = offs.shape[0]
            strided = out_set is not self and out_set.stride == 2 * self.stride and offset_stride == self.stride
            par = self._gen_parent
            # tables that their builder pre-fills with -1 come from the static trace's pre-cleared arena when there is one
            prefilled = method == "auto" and out_set.n > 0 and self.n > 0 and not (
                out_set is self and kernel_size == 3 and par is not None and offset_stride == self.stride and self.n == 8 * par.n) and (
                (out_set is self and kernel_size % 2 == 1 and kernel_size > 1) or (strided and kernel_size in (1, 2, 3)))
            plan = P.current()
            a = plan.arena_ff(out_set.n * K * 4, self.device) if (prefilled and plan is not None) else None
            nbr = a[:out_set.n * K * 4].view(torch.int32).view(out_set.n, K) if a is not None else \
                torch.empty((out_set.n, K), dtype=torch.int32, device=self.device)
            pre = int(a is not None)
            if out_set.n == 0 or self.n == 0:
                nbr.fill_(-1)
            elif method == "auto" and out_set is self and kernel_size == 3 and par is not None and \
                    offset_stride == self.stride and self.n == 8 * par.n:
                # a generated child set: its table follows from the parents' own 3x3x3 table (no hash map of this set at all)
                call("cnrma_sparse_kernel_map_children", ptr(par.neighbours(par, 3, par.stride)), par.n, ptr(par.n_dev), ptr(nbr),
                     stream())
            elif method == "auto" and out_set is self and kernel_size % 2 == 1 and kernel_size > 1:
                m = self.cmap
                call("cnrma_sparse_kernel_map_symmetric", ptr(self.C), self.n, ptr(self.n_dev), ptr(m.keys), ptr(m.vals), m.cap,
                     ptr(offs), K, ptr(nbr), pre, stream())
            elif method == "auto" and strided and kernel_size in (1, 2, 3):
                m = out_set.cmap
                call("cnrma_sparse_kernel_map_strided", ptr(self.C), self.n, ptr(self.n_dev), self.stride, kernel_size,
                     ptr(m.keys), ptr(m.vals), m.cap, ptr(nbr), out_set.n, pre, stream())
            else:
                m = self.cmap
                call("cnrma_sparse_kernel_map", ptr(out_set.C), out_set.n, ptr(out_set.n_dev), ptr(m.keys), ptr(m.vals), m.cap,
                     ptr(offs), K, ptr(nbr), stream())
            self._nbr[key] = (nbr, out_set)   # keep out_set alive so that id() stays unique
        return self._nbr[key][0]


def tile_union(in_cs, out_set, kernel_size, offset_stride):
    """per 64-row output tile: the distinct input rows its 27 offsets read + local indices (cnrma_sparse_tile_union_build);
    cached on the input coordinate set next to the neighbour table it is derived from"""
    key = (kernel_size, offset_stride, id(out_set))
    tu = in_cs._union.get(key)
    if tu is None:
        nbr = in_cs.neighbours(out_set, kernel_size, offset_stride)
        tu = torch.empty(_lib.load().cnrma_sparse_tile_union_bytes(out_set.n), dtype=torch.uint8, device=in_cs.device)
        call("cnrma_sparse_tile_union_build", ptr(nbr), out_set.n, ptr(out_set.n_dev), nbr.shape[1], ptr(tu), stream())
        in_cs._union[key] = tu
    return tu


class SparseTensor:
    """features F [N,C] fp32 on a CoordSet; mirrors the attributes of ME.SparseTensor the reference touches."""

    def __init__(self, features, coord_set, split=None, amax=None):
        assert features.shape[0] == coord_set.n
        self.F = features
        self.cs = coord_set
        self._split = split       # bf16 [n+1, C/8, 3, 8] companion (hi/mid/lo pieces), filled by conv epilogues
        self.amax = amax          # device scalar: an upper bound of |F| (f16x3 convolutions scale their operands by it)

    def absmax(self):
        """upper bound of |F| as a device scalar: the producing kernel's running maximum, or one pass over F"""
        if self.amax is None:
            n, C = self.F.shape
            # the entry point zeroes the slots itself; an empty tensor's bound is 0
            out = (torch.empty if n else torch.zeros)(_AMAX_WORDS, dtype=torch.float32, device=self.F.device)
            if n:
                call("cnrma_absmax_f32", ptr(self.F.contiguous()), n, ptr(self.cs.n_dev), C, ptr(out), stream())
            self.amax = out
        return self.amax

    def split(self):
        """pre-split bf16 companion of the features for the bf16x6 convolutions (built on demand, cached)"""
        if self._split is None:
            n, C = self.F.shape
            sp = torch.empty((n + 1, C // 8, 3, 8), dtype=torch.bfloat16, device=self.F.device)
            call("cnrma_sparse_split_features", ptr(self.F.contiguous()), max(n, 1), ptr(self.cs.n_dev), C, ptr(sp), stream()) if n else sp.zero_()
            self._split = sp
        return self._split

    @property
    def C(self):
        return self.cs.C

    @property
    def features(self):
        return self.F

    @property
    def coordinates(self):
        return self.cs.C

    @property
    def tensor_stride(self):
        return [self.cs.stride] * 3

    @property
    def device(self):
        return self.F.device

    def __len__(self):
        return self.cs.n

    @property
    def decomposition_permutations(self):
        if self.cs.n_batch <= 1:
            return [torch.arange(self.cs.n, device=self.device)]
        b = self.cs.C[:, 0]
        return [torch.nonzero(b == i).squeeze(1) for i in range(self.cs.n_batch)]

    @property
    def decomposed_coordinates(self):
        return [self.cs.C[p, 1:] for p in self.decomposition_permutations]

    def features_at_coordinates(self, query_coords_float, n_dev=None):
        q = query_coords_float.to(torch.int32).contiguous()
        return interpolate(self, q, n_dev)

    def __add__(self, other):
        return union_add(self, other)


# --------------------------------------------------------------------------------------------------------------
def _voxelize_enqueue(coords, feats, voxel_size, batch_id, row_order, m_dev=None, out_cap=0, out=None):
    coords = coords.contiguous().float()
    feats = feats.contiguous().float()
    M, C = feats.shape
    dev = feats.device
    m = CoordMap(M, dev)
    if out is not None:                # caller-owned row blocks [M,4] / [M,C] (slices of a multi-scene buffer)
        out_c, out_f = out
        assert out_c.shape == (M, 4) and out_f.shape == (M, C) and out_c.is_contiguous() and out_f.is_contiguous()
    else:
        out_c = torch.empty((M, 4), dtype=torch.int32, device=dev)
        out_f = torch.empty((M, C), dtype=torch.float32, device=dev)
    src = torch.empty(M, dtype=torch.int32, device=dev)
    n_out = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(_lib.load().cnrma_voxelize_workspace_bytes(M), dtype=torch.uint8, device=dev)
    call("cnrma_voxelize_f32", ptr(coords), ptr(feats), M, ptr(m_dev), C, float(voxel_size), int(batch_id),
         {"first": 0, "morton": 1}[row_order], ptr(m.keys), ptr(m.vals),
         m.cap, ptr(out_c), ptr(out_f), ptr(src), out_cap, ptr(n_out), ptr(ws), stream())
    return out_c, out_f, src, n_out, m


def voxelize(coords, feats, voxel_size, batch_id=0, row_order="morton", n_dev=None):
    """ME.utils.batch_sparse_collate + ME.SparseTensor (ray_marching.py:328-330): floor(coord / voxel_size),
    first occurrence wins.  row_order "first" = rows in first-occurrence order, "morton" = rows sorted by the
    Morton code of the voxel (ME's own order is implementation-defined).  n_dev: device word with the live number of
    input rows (static trace).  Returns (SparseTensor @ stride 1, src_index int32 = source row of every output row)."""
    _lib.require_gpu()
    if P.static():
        plan = P.current()
        cap = plan.next_cap(coords.shape[0])
        out_c, out_f, src, n_out, m = _voxelize_enqueue(coords, feats, voxel_size, batch_id, row_order, n_dev, cap)
        plan.watch(n_out, 1, cap)
        cs = CoordSet(out_c, 1, m, n=cap, n_dev=n_out)
        cs.compact = cs.sorted = row_order == "morton"
        return SparseTensor(out_f[:cap], cs), src[:cap]
    out_c, out_f, src, n_out, m = _voxelize_enqueue(coords, feats, voxel_size, batch_id, row_order, n_dev)
    n = _count(n_out, coords.shape[0])[0]
    cs = CoordSet(out_c[:n], 1, m)
    cs.compact = cs.sorted = row_order == "morton"
    if _train(feats):                     # training: the surviving rows through torch indexing (keeps the graph)
        return SparseTensor(feats.index_select(0, src[:n].long()), cs), src[:n]
    return SparseTensor(out_f[:n], cs), src[:n]


def sparse_collate(list_of_coords_feats, voxel_size):
    """Multi-scene variant: one SparseTensor holding all scenes (batch id = list index)."""
    if len(list_of_coords_feats) == 1:
        return voxelize(*list_of_coords_feats[0], voxel_size, 0)[0]
    _lib.require_gpu()
    if P.static():
        raise _lib.CnrmaError("static trace: use sparse_collate_static (device-side row counts)")
    parts = [_voxelize_enqueue(c, f, voxel_size, b, "morton") for b, (c, f) in enumerate(list_of_coords_feats)]
    counts = _lib.read_ints(torch.cat([p[3] for p in parts]))       # ONE device->host read for all scenes
    C = torch.cat([p[0][:n] for p, n in zip(parts, counts)])
    if _train(*[f for _, f in list_of_coords_feats]):
        F = torch.cat([f.float().index_select(0, p[2][:n].long()) for (_, f), p, n in zip(list_of_coords_feats, parts, counts)])
    else:
        F = torch.cat([p[1][:n] for p, n in zip(parts, counts)])
    cs = CoordSet(C, 1, None, len(parts))
    cs.scene_major = True
    cs.compact = cs.sorted = True             # scene blocks in batch order, each in Morton order: the key carries the batch on top
    cs._counts = counts
    return SparseTensor(F, cs)


def sparse_collate_static(list_of_coords_feats_ndev, voxel_size):
    """sparse_collate inside the static trace: scene b's points are rows [0, n_dev_b) of capacity-sized (coords, feats).
    Every scene is voxelised into its own block of one [sum(cap_b), .] buffer; the blocks' live rows are then packed to
    the front by ONE gather whose row map is computed on the device from the B row counts (cumsum + searchsorted) --
    scene-major rows, batch id = list index, exactly what the eager sparse_collate builds, without a read-back."""
    _lib.require_gpu()
    plan = P.current()
    assert plan is not None and plan.static
    B = len(list_of_coords_feats_ndev)
    dev = list_of_coords_feats_ndev[0][1].device
    C = list_of_coords_feats_ndev[0][1].shape[1]
    caps = [plan.next_cap(c.shape[0]) for c, _, _ in list_of_coords_feats_ndev]
    in_caps = [c.shape[0] for c, _, _ in list_of_coords_feats_ndev]
    # the voxeliser writes at most `cap` rows but needs input-sized row blocks: place the blocks at input-size pitch
    base = [0]
    for m_in in in_caps:
        base.append(base[-1] + m_in)
    big_c = torch.empty((base[-1], 4), dtype=torch.int32, device=dev)
    big_f = torch.empty((base[-1], C), dtype=torch.float32, device=dev)
    counts = []
    for b, (c, f, n_dev) in enumerate(list_of_coords_feats_ndev):
        blk = (big_c[base[b]:base[b + 1]], big_f[base[b]:base[b + 1]])
        _, _, _, n_out, _ = _voxelize_enqueue(c, f, voxel_size, b, "morton", n_dev, caps[b], out=blk)
        plan.watch(n_out, 1, caps[b])
        counts.append(n_out)
    cap_total = sum(caps)
    n = torch.cat(counts)                                              # [B] live rows per scene
    incl = torch.cumsum(n, 0)
    excl = (incl - n)
    r = plan.const(lambda: torch.arange(cap_total, dtype=torch.int64, device=dev))
    bases = plan.const(lambda: torch.tensor(base[:B], dtype=torch.int64, device=dev))
    sb = torch.searchsorted(incl.to(torch.int64), r, right=True).clamp_(max=B - 1)      # scene of output row r
    src = (bases[sb] + (r - excl.to(torch.int64)[sb])).clamp_(0, base[-1] - 1)          # rows >= total: any valid row
    total = incl[B - 1:].to(torch.int32).contiguous()
    cs = CoordSet(big_c.index_select(0, src), 1, None, B, n=cap_total, n_dev=total)
    cs.scene_major = True
    cs.compact = cs.sorted = True
    cs._counts_dev = (n.to(torch.int32).contiguous(), excl.to(torch.int32).contiguous())
    return SparseTensor(big_f.index_select(0, src), cs)


def fold_bn(bn, bias=None):
    """eval-mode BatchNorm1d (+ optional conv bias) -> per-channel (scale, shift)."""
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    shift = bn.bias - bn.running_mean * scale
    if bias is not None:
        shift = shift + bias.view(-1) * scale
    return scale.contiguous().float(), shift.contiguous().float()


ACT = {None: 0, "none": 0, "relu": 1, "elu": 2}

_WS = {}
_WS_STREAMS = 12      # scratch buffers kept: the streams of a process come and go (every scene slot / detector has its own), their
                      # raw handles are all this table knows of them -- the least recently used entries are dropped (round 6: ~1 GB
                      # per detector ever built stayed allocated).  Dropping is safe: the caching allocator hands a freed block
                      # back to the stream it was allocated on, in stream order.


def _lru_get(table, key, make, limit=_WS_STREAMS):
    """table[key] (moved to the most-recent end), created by make() when absent; the oldest entries beyond `limit` are dropped"""
    buf = table.pop(key, None)
    if buf is None:
        buf = make()
    table[key] = buf
    while len(table) > limit:
        table.pop(next(iter(table)))
    return buf


def _workspace(nbytes, device):
    """grow-only scratch buffer per (device, stream): reuse is stream-ordered, so concurrent streams (several scenes in
    flight on one GPU) must not share it"""
    if P.static():
        return P.current().workspace(nbytes, device)
    key = (device, stream())
    buf = _WS.pop(key, None)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    return _lru_get(_WS, key, lambda: buf)


_COUNTER_WORDS = 16384
_COUNTERS = {}
# split gather-once convolutions: the last block of a tile adds the partial slabs up itself instead of a reduce launch behind
# the kernel.  Built and measured: 2.5-3.5x SLOWER (11 k rows x 128 -> 128: 48 -> 120-172 us; 2.4 k x 256 -> 256: 44 -> 140-174):
# the partial tiles cross XCDs inside one kernel, and the device-scope release / acquire fences around the counter write back
# and invalidate the XCD's L2 for every block.  Off; the reduce launch (9 us) stays.
GO_INKERNEL_REDUCE = False


def _tile_counters(device):
    """zeroed arrival counters of the split gather-once convolutions (the last block of a tile reduces the slabs and resets
    its word): per plan inside a static trace, per (device, stream) otherwise"""
    if P.static():
        return P.current().counters(device, _COUNTER_WORDS)
    key = (device, stream())
    return _lru_get(_COUNTERS, key, lambda: torch.zeros(_COUNTER_WORDS, dtype=torch.int32, device=device))


# "f16x3":  22-bit operands on the fp16 matrix cores: a 2^s = h + m (two fp16 pieces, power-of-two scale from the tensor's
#           magnitude), products hh + hm + mh in fp32 -- relative error <= 3 x 2^-22 per product, below the rounding noise
#           of an fp32 accumulation of the same length; half the matrix work and 2/3 of the LDS traffic of "bf16x6";
# "bf16x6": 24-bit operands on the bf16 matrix cores (3-way exact operand split, 6 partial products);
# "f32":    v_mfma_f32_32x32x2_f32 (bit-for-bit an fp32 fma chain).  Layers with Cin % 32 != 0 always use "f32".
CONV_PRECISION = "f16x3"
PRESPLIT = False     # carry bf16 hi/mid/lo companions of the features between convolutions (see conv())


_tls = threading.local()
_AMAX_WORDS = 64 * 16          # cnrma_amax_bytes() / 4


def _amax_slot(device):
    """a zeroed magnitude bound (cnrma_amax_bytes: 64 words, one per 64-byte line) for a kernel's running |output|
    maximum: bounds are carved from a chunk of 64 that is zeroed once (one memset per 64 convolutions) and never reused,
    so a tensor's bound stays valid as long as it lives"""
    if P.static():                   # inside the trace: the chunk's memset is part of the captured sequence
        return P.current().amax_slot(device, _AMAX_WORDS)
    pool = getattr(_tls, "amax_pool", None)
    key = (device, stream())
    if pool is None or pool[0] != key or pool[2] >= 64:
        pool = [key, torch.zeros(64 * _AMAX_WORDS, dtype=torch.float32, device=device), 0]
        _tls.amax_pool = pool
    i = pool[2]
    pool[2] = i + 1
    return pool[1][i * _AMAX_WORDS:(i + 1) * _AMAX_WORDS]


# Prepared weight images (fp16 / bf16 splits) are cached per weight tensor in a weak table -- nothing is attached to the
# Parameter itself, so pickling a module never drags GPU buffers along -- keyed on (_version, data_ptr, device).  Writes
# through `param.data` (EMA hooks, manual .data.copy_()) change neither: call invalidate_weight_cache() after them;
# load_state_dict() and Module.train() of the sparse modules do so themselves (cnrma_amd.nn).
_WEIGHT_CACHE = WeakIdKeyDictionary()          # identity-keyed: Tensor.__eq__ is elementwise


def _cache_get(weight, kind):
    d = _WEIGHT_CACHE.get(weight)
    return None if d is None else d.get(kind)


def _cache_put(weight, kind, value):
    try:
        _WEIGHT_CACHE.setdefault(weight, {})[kind] = value
    except TypeError:
        pass


def invalidate_weight_cache(module_or_tensor=None):
    """drop the prepared weight images of one tensor, of every parameter of a module, or (None) of everything"""
    if module_or_tensor is None:
        _WEIGHT_CACHE.clear()
    elif isinstance(module_or_tensor, torch.nn.Module):
        for p_ in module_or_tensor.parameters():
            _WEIGHT_CACHE.pop(p_, None)
    else:
        _WEIGHT_CACHE.pop(module_or_tensor, None)


def _pinned(ws):
    """inside a static trace the prepared image is pinned to the plan: the graph replays its raw pointer, while the weak
    cache drops the image on train() / load_state_dict() / when the weight tensor dies"""
    if P.static():
        P.current().keep(ws)
    return ws


def split_weights_f16(weight):
    """weight fp32 [K,Cin,Cout] (or [Cin,Cout]) -> fp16 [2,K,Cout,Cin] (hi / lo pieces of weight * 2^s) + a trailer word
    holding max|weight| (the kernel derives s from it).  Cached on the weight tensor like split_weights()."""
    tag = (weight._version, weight.data_ptr(), weight.device)
    hit = _cache_get(weight, "_cnrma_split_f16")
    if hit is not None and hit[0] == tag:
        return _pinned(hit[1])
    w = weight.detach().contiguous().float()
    if w.dim() == 2:
        w = w.unsqueeze(0)
    K, Cin, Cout = w.shape
    ws = torch.empty(_lib.load().cnrma_sparse_conv_f16_weight_bytes(K, Cin, Cout), dtype=torch.uint8, device=w.device)
    call("cnrma_sparse_conv_prepare_weights_f16", ptr(w), K, Cin, Cout, ptr(ws), stream())
    _cache_put(weight, "_cnrma_split_f16", (tag, ws))
    return _pinned(ws)


def split_weights_f16_frag(weight):
    """weight fp32 [27,Cin,Cout] -> the fp16 hi / lo image in MFMA-fragment order (gather-once convolution), cached like
    split_weights_f16()"""
    tag = (weight._version, weight.data_ptr(), weight.device)
    hit = _cache_get(weight, "_cnrma_frag_f16")
    if hit is not None and hit[0] == tag:
        return _pinned(hit[1])
    w = weight.detach().contiguous().float()
    K, Cin, Cout = w.shape
    ws = torch.empty(_lib.load().cnrma_sparse_conv_f16_weight_bytes(K, Cin, Cout), dtype=torch.uint8, device=w.device)
    call("cnrma_sparse_conv_prepare_weights_f16_frag", ptr(w), K, Cin, Cout, ptr(ws), stream())
    _cache_put(weight, "_cnrma_frag_f16", (tag, ws))
    return _pinned(ws)


def weights_f32_frag(weight):
    """weight fp32 [27,Cin,Cout] -> the fp32 image in MFMA-fragment order (exact-fp32 gather-once convolution), cached like
    split_weights_f16()"""
    tag = (weight._version, weight.data_ptr(), weight.device)
    hit = _cache_get(weight, "_cnrma_frag_f32")
    if hit is not None and hit[0] == tag:
        return _pinned(hit[1])
    w = weight.detach().contiguous().float()
    K, Cin, Cout = w.shape
    ws = torch.empty(_lib.load().cnrma_sparse_conv_f32_frag_weight_bytes(K, Cin, Cout), dtype=torch.uint8, device=w.device)
    call("cnrma_sparse_conv_prepare_weights_f32_frag", ptr(w), K, Cin, Cout, ptr(ws), stream())
    _cache_put(weight, "_cnrma_frag_f32", (tag, ws))
    return _pinned(ws)


def weights_bf16(weight):
    """weight fp32 [K,Cin,Cout] (or [Cin,Cout]) -> bf16 [K,Cout_p,Cin] (round to nearest) for the "bf16" convolutions;
    cached on the weight tensor like the other prepared images"""
    tag = (weight._version, weight.data_ptr(), weight.device)
    hit = _cache_get(weight, "_cnrma_bf16")
    if hit is not None and hit[0] == tag:
        return _pinned(hit[1])
    w = weight.detach().contiguous().float()
    if w.dim() == 2:
        w = w.unsqueeze(0)
    K, Cin, Cout = w.shape
    ws = torch.empty(_lib.load().cnrma_sparse_conv_bf16_weight_bytes(K, Cin, Cout), dtype=torch.uint8, device=w.device)
    call("cnrma_sparse_conv_prepare_weights_bf16", ptr(w), K, Cin, Cout, ptr(ws), stream())
    _cache_put(weight, "_cnrma_bf16", (tag, ws))
    return _pinned(ws)


def weights_bf16_frag(weight, transposed=False):
    """weight fp32 [27,Cin,Cout] -> the bf16 MFMA-fragment image of cnrma_sparse_conv_go_bf16 (transposed: the mirrored +
    transposed image of the data gradient); cached on the weight tensor.  When both widths allow it the two images are made
    by ONE launch at the first request of a step -- the forward's, whose host-bound launch stream has room for it"""
    tag = (weight._version, weight.data_ptr(), weight.device)
    hit = _cache_get(weight, "_cnrma_bf16_frag")
    if hit is not None and hit[0] == tag and hit[1 + int(transposed)] is not None:
        return _pinned(hit[1 + int(transposed)])
    w = weight.detach().contiguous().float()
    K, Cin, Cout = w.shape
    lib = _lib.load()
    fwd = hit[1] if hit is not None and hit[0] == tag else None
    tr = hit[2] if hit is not None and hit[0] == tag else None
    if Cin % 32 == 0 and Cout % 32 == 0 and Cin >= 64 and fwd is None and tr is None:
        fwd = torch.empty(lib.cnrma_sparse_conv_bf16_frag_weight_bytes(K, Cin, Cout), dtype=torch.uint8, device=w.device)
        tr = torch.empty(lib.cnrma_sparse_conv_bf16_frag_weight_bytes(K, Cout, Cin), dtype=torch.uint8, device=w.device)
        call("cnrma_sparse_conv_prepare_weights_bf16_frag_pair", ptr(w), K, Cin, Cout, 1, ptr(fwd), ptr(tr), stream())
    elif transposed:
        tr = torch.empty(lib.cnrma_sparse_conv_bf16_frag_weight_bytes(K, Cout, Cin), dtype=torch.uint8, device=w.device)
        call("cnrma_sparse_conv_prepare_weights_bf16_frag", ptr(w), K, Cin, Cout, 1, 1, ptr(tr), stream())
    else:
        fwd = torch.empty(lib.cnrma_sparse_conv_bf16_frag_weight_bytes(K, Cin, Cout), dtype=torch.uint8, device=w.device)
        call("cnrma_sparse_conv_prepare_weights_bf16_frag", ptr(w), K, Cin, Cout, 0, 0, ptr(fwd), stream())
    _cache_put(weight, "_cnrma_bf16_frag", (tag, fwd, tr))
    return _pinned(tr if transposed else fwd)


def _precision(precision=None):
    """explicit > torch.autocast(bf16) region (the reference's training configuration: bf16 operands, fp32 accumulation,
    BASELINE configs[4]) > the module default CONV_PRECISION"""
    if precision:
        return precision
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return "bf16"
    return CONV_PRECISION


def split_weights(weight):
    """weight fp32 [K,Cin,Cout] (or [Cin,Cout]) -> bf16 [3,K,Cout_p,Cin] (hi / mid / lo pieces, Cout padded to 128).  The result is cached ON
    the weight tensor object (so it dies with it) and rebuilt when the tensor is modified in place or moved."""
    tag = (weight._version, weight.data_ptr(), weight.device)
    hit = _cache_get(weight, "_cnrma_split")
    if hit is not None and hit[0] == tag:
        return _pinned(hit[1])
    w = weight.detach().contiguous().float()
    if w.dim() == 2:
        w = w.unsqueeze(0)
    K, Cin, Cout = w.shape
    ws = torch.empty(_lib.load().cnrma_sparse_conv_weight_bytes(K, Cin, Cout), dtype=torch.uint8, device=w.device)
    call("cnrma_sparse_conv_prepare_weights", ptr(w), K, Cin, Cout, ptr(ws), stream())
    _cache_put(weight, "_cnrma_split", (tag, ws))
    return _pinned(ws)


CONV_MODES = {"f32": 0, "f16x3": 1, "bf16": 2, "bf16x6": 3}
CONV_SHAPES = ("128x128", "128x64", "64x64", "128x32", "64x128", "256x128", "256x64")


def conv_plan(n_out, Cin, Cout, K, precision=None, slices=1):
    """which kernel variant conv() launches for an output capacity of n_out rows: dict(tile=(rows, cols), splits,
    k_per_split, prefetch, shape) -- cnrma_sparse_conv_plan, a pure host function (tests prove variant coverage with it)"""
    import ctypes
    prec = precision or CONV_PRECISION
    mode = CONV_MODES[prec] if Cin % 32 == 0 else 0
    ws = _lib.load().cnrma_sparse_conv_workspace_bytes(n_out, Cout, K) if slices == 1 else 0
    out = (ctypes.c_int * 6)()
    call("cnrma_sparse_conv_plan", int(n_out), int(Cin), int(Cout), int(K), mode, int(slices), ws, out)
    return dict(tile=(out[0], out[1]), splits=out[2], k_per_split=out[3], prefetch=out[4], shape=CONV_SHAPES[out[5]])


def conv_go_plan(n_out, Cin, Cout, residual=False):
    """which gather-once kernel variant conv() launches for an output capacity of n_out rows (3x3x3, stride 1, f16x3):
    dict(form, columns, ks, splits, slices_per_split, order, residual_in_kernel, blocks, offsets_in_flight, workspace) --
    cnrma_sparse_conv_go_plan, a pure host function"""
    import ctypes
    ws = n_out * Cout * 4 * (Cin // 32) if n_out < GO_WS_ROWS else 0
    out = (ctypes.c_int * 8)()
    call("cnrma_sparse_conv_go_plan", int(n_out), int(Cin), int(Cout), ws, int(bool(residual)), out)
    return dict(form=out[0], columns=out[1], ks=2 if out[1] == 64 else 1, splits=out[2], slices_per_split=out[3],
                order=("plain", "groups->xcd", "tiles->xcd")[out[4]], residual_in_kernel=bool(out[5]), blocks=out[6],
                offsets_in_flight=out[7], workspace=ws)


def conv_tuning(shape=None, splits=-1, pf=-1, ablate=0, ws=-1, xcd=-1, go=-1, nb=-1):
    """debug / A-B aid (scripts/conv_sweep.py, variant-forcing tests): force the tile shape ("128x128", ...), the split count
    and the prefetch depth of every later convolution launch; no arguments = the product configuration"""
    import ctypes
    if shape is None and splits < 0 and pf < 0 and not ablate and ws < 0 and xcd < 0 and go < 0 and nb < 0:
        # back to the product library (libcnrma_hip.so has no tuning state, nothing to reset there)
        if _lib.experiments_active():
            call("cnrma_debug_conv_tuning", None, 0)
            _lib.experiments(False, "conv")
        return
    _lib.experiments(True, "conv")       # the forced variants / alternative kernel forms exist in libcnrma_hip_exp.so only
    # ablate (diagnostic kernels, timing only -- results are wrong): bit 0 no MFMAs, 1 no A loads, 2 no B loads, 3 no LDS
    # stores, 4 no barriers after a block's first stage; the gather-once weight gradient reads 256 no consumer phase, 512 no
    # LDS stores, 1024 no row loads, 2048 LDS reads without MFMAs (scripts/wgrad_go_ablate.py)
    # ws: LDS ring slots of the warp-specialised f16x3 kernel (2..4; 0 = the stage kernel; -1 = the launcher's default)
    # xcd: stage kernel: 1 = each XCD works on one contiguous eighth of the row tiles (0 = tiles dealt round-robin);
    #      gather-once second form: 0 plain order, 1 (column tile, slice split) groups -> XCDs, 2 row tiles -> XCDs
    # go: gather-once kernel form (0 = first form, 1 / 2 = second form with 1 / 2 row tiles per block); nb: weight offsets in
    #     flight per wave of the second form (2 / 4)
    arr = (ctypes.c_int * 8)(CONV_SHAPES.index(shape) if shape is not None else -1, int(splits), int(pf), int(ablate), int(ws),
                             int(xcd), int(go), int(nb))
    call("cnrma_debug_conv_tuning", arr, 8)


GO_CONV = "auto"     # gather-once kernel for the 3x3x3 stride-1 convolutions in f16x3 (csrc/sparse.hip): "auto" = on coordinate sets
                     # whose rows are compact (CoordSet.compact) with >= GO_MIN_ROWS rows; True / False force it (tests, A/B runs)
GO_UMAX = 280        # csrc/sparse.hip GO_UMAX: rows of a tile's union image in LDS (the local index of "no neighbour")
GO_WS_ROWS = 65536   # below: a workspace for the split over channel slices is handed to the kernel (it decides)
GO_F32 = True        # CONV_PRECISION = "f32": the exact-fp32 gather-once kernel where the f16x3 one would run (False: the stage kernel)
GO_STAMPS = None     # diagnostic build of the second form (conv_tuning(ablate=64)): an int64 tensor of 16 words per block
GO_MIN_ROWS = 256    # below: the stage kernel split over the 27 offsets.  (Round 4: 1024 -- the first form ran the 541-row level at
                     # 0.95x of the stage kernel; the second form with its (column tile, slice) groups pinned to XCDs runs it at 35 us
                     # against 56: profiles/r05_go_forms2_S.log)
PAIR_HDR_BYTES, PAIR_OVERFLOW_WORD = 512, 64 + 34      # csrc/sparse.hip: PAIR_HDR ints, hdr[64 + 34] = "an entry was dropped"
PAIR_CONV = True     # pair-list kernel for stride-2 convolutions whose kernel map is nearly empty (the stem)
# regrouping the table costs ~0.2 ms per 450 k output rows (count, plan, fill, reduce: MI355X); the tile kernel wastes
# 27 x Cin / 32 stages per tile on empty offsets -- measured break-even between Cin = 32 (tile kernel 0.18 ms, pair list
# 0.29 ms) and Cin = 256 (1.23 ms vs 0.45 ms)
PAIR_CONV_MIN_CIN = 128


def _gather_once(in_cs, out_cs):
    """3x3x3 stride-1 convolution: the gather-once kernel where the rows are compact (a 64-row tile reads ~250 distinct
    input rows instead of 64 x ~20) and there are enough of them to fill the chip; GO_CONV True / False force the choice"""
    if GO_CONV == "auto":
        if not in_cs.compact:
            return False
        # the row threshold is a recorded branch of the size plan: out_cs.n is a live count in the eager /
        # calibration run and a CAPACITY (recorded size x 1.2 + 256, i.e. always >= GO_MIN_ROWS) in the static trace -- decided on
        # `n` alone a 9-56-row level ran the 64-row-tile gather-once kernel in the replay and the stage kernel in the eager
        # fallback of the same scene (another summation order between the two)
        p = P.current()
        if p is not None and p.static:
            f = p.next_flag()
            return out_cs.n >= GO_MIN_ROWS if f is None else bool(f)      # calibration scenes disagreed: by capacity
        f = out_cs.n >= GO_MIN_ROWS
        if p is not None:
            p.record_flag(f)
        return f
    return bool(GO_CONV)


def _nearly_empty_map(in_cs, out_cs):
    """stride-2 convolution: an input row is a neighbour of <= 2 outputs per axis (27 / 8 on average), so the table is
    at most 3.4 N_in / (27 N_out) full.  True when that bound is below 20 % -- the coarsening merged almost nothing, i.e.
    the input is a sparse point sample.  A recorded branch of the size plan (the static trace has capacities, not counts)."""
    p = P.current()
    if p is not None and p.static:
        return bool(p.next_flag())                       # None (calibration scenes disagreed) -> the tile kernel
    f = 3.375 * in_cs.n < 0.2 * 27 * out_cs.n
    if p is not None:
        p.record_flag(f)
    return f


def conv(x, weight, kernel_size=3, stride=1, scale=None, shift=None, residual=None, act=None, precision=None):
    """MinkowskiConvolution + fused epilogue: out = act((sum_k in[nbr] @ W[k]) * scale + shift + residual).
    weight [K,Cin,Cout] (or [Cin,Cout] when K == 1)."""
    _lib.require_gpu()
    w = weight.contiguous().float()
    if w.dim() == 2:
        w = w.unsqueeze(0)
    K, Cin, Cout = w.shape
    assert K == kernel_size ** 3 and Cin == x.F.shape[1]
    in_cs = x.cs
    out_cs = in_cs if stride == 1 else in_cs.strided(stride)
    if kernel_size == 1 and stride == 1:
        nbr = None
    else:
        nbr = in_cs.neighbours(out_cs, kernel_size, in_cs.stride)
    out = torch.empty((out_cs.n, Cout), dtype=torch.float32, device=x.device)
    out_split = None
    if out_cs.n:
        res = residual.F.contiguous() if isinstance(residual, SparseTensor) else residual
        if res is not None:
            assert res.shape == out.shape
        ws_bytes = _lib.load().cnrma_sparse_conv_workspace_bytes(out_cs.n, Cout, K)
        ws = _workspace(ws_bytes, x.device) if ws_bytes else None
        prec = _precision(precision)
        if P.static():       # the captured launches keep raw pointers: pin what the modules' caches may drop
            P.current().keep(weight, w, scale, shift)
        if prec == "bf16" and Cin % 32 == 0:
            call("cnrma_sparse_conv_bf16", ptr(x.F.contiguous()), Cin, ptr(nbr), K, ptr(weights_bf16(weight)), Cout, ptr(scale),
                 ptr(shift), ptr(res), ACT[act], ptr(out), out_cs.n, ptr(out_cs.n_dev), ptr(ws), ws_bytes, stream())
            return SparseTensor(out, out_cs)
        if prec == "f16x3" and Cin % 32 == 0:
            out_amax = _amax_slot(x.device)
            if PAIR_CONV and stride == 2 and K == 27 and Cout % 4 == 0 and Cin >= PAIR_CONV_MIN_CIN and _nearly_empty_map(in_cs, out_cs):
                # pair-list kernel: the kernel map of a stride-2 convolution on a point sample is ~5 % full
                pair_cap = (min(27 * out_cs.n, 8 * in_cs.n) + 128 * K + 127) // 128 * 128
                pw_bytes = _lib.load().cnrma_sparse_conv_pairs_workspace_bytes(out_cs.n, K, Cout, pair_cap)
                pw = _workspace(pw_bytes, x.device)
                call("cnrma_sparse_conv_pairs_f16x3", ptr(x.F.contiguous()), ptr(x.absmax()), Cin, ptr(nbr), K,
                     ptr(split_weights_f16(weight)), Cout, ptr(scale), ptr(shift), ptr(res), ACT[act], ptr(out), ptr(out_amax),
                     out_cs.n, ptr(out_cs.n_dev), pair_cap, ptr(pw), pw_bytes, stream())
                if P.static():
                    # the kernels flag pair entries dropped for want of capacity in word 98 of the workspace header (the
                    # capacity above is the provable bound, so this never fires; a violated assumption must still never be
                    # silent: the word joins the plan's status -- copied out, the workspace is reused by the next layer)
                    P.current().watch(pw[:PAIR_HDR_BYTES].view(torch.int32)[PAIR_OVERFLOW_WORD:PAIR_OVERFLOW_WORD + 1].clone(), 0, 0)
                return SparseTensor(out, out_cs, None, out_amax)
            if K == 27 and stride == 1 and Cout >= 64 and _gather_once(in_cs, out_cs):
                # gather-once kernel: a tile's distinct input rows staged once per channel slice, offsets run from LDS
                go_ws_bytes = out_cs.n * Cout * 4 * (Cin // 32) if out_cs.n < GO_WS_ROWS else 0
                go_ws = _workspace(go_ws_bytes, x.device) if go_ws_bytes else None
                counters = _tile_counters(x.device) if GO_INKERNEL_REDUCE and go_ws_bytes and \
                    -(-out_cs.n // 64) * -(-Cout // 64) <= _COUNTER_WORDS else GO_STAMPS
                call("cnrma_sparse_conv_go_f16x3", ptr(x.F.contiguous()), ptr(x.absmax()), Cin,
                     ptr(tile_union(in_cs, out_cs, kernel_size, in_cs.stride)), ptr(split_weights_f16_frag(weight)), Cout,
                     ptr(scale), ptr(shift), ptr(res), ACT[act], ptr(out), ptr(out_amax), out_cs.n, ptr(out_cs.n_dev),
                     ptr(go_ws), go_ws_bytes, ptr(counters), stream())
                return SparseTensor(out, out_cs, None, out_amax)
            call("cnrma_sparse_conv_f16x3", ptr(x.F.contiguous()), ptr(x.absmax()), Cin, ptr(nbr), K,
                 ptr(split_weights_f16(weight)), Cout, ptr(scale), ptr(shift), ptr(res), ACT[act], ptr(out), ptr(out_amax),
                 out_cs.n, ptr(out_cs.n_dev), ptr(ws), ws_bytes, stream())
            return SparseTensor(out, out_cs, None, out_amax)
        if prec == "bf16x6" and Cin % 32 == 0:
            # pre-split companions (PRESPLIT): measured on MI355X at the ScanNet shape they do not pay -- the kernel is
            # bound by L2->LDS gather traffic, not by the in-loop split (5.6 ms either way) -- so they are off by default
            in_split = x.split() if (PRESPLIT or x._split is not None) else None
            if PRESPLIT and Cout % 8 == 0:
                out_split = torch.empty((out_cs.n + 1, Cout // 8, 3, 8), dtype=torch.bfloat16, device=x.device)
            call("cnrma_sparse_conv_bf16x6", ptr(x.F.contiguous()), ptr(in_split), x.cs.n, Cin, ptr(nbr), K, ptr(split_weights(weight)),
                 Cout, ptr(scale), ptr(shift), ptr(res), ACT[act], ptr(out), ptr(out_split), out_cs.n, ptr(out_cs.n_dev), ptr(ws),
                 ws_bytes, stream())
        elif prec == "f32" and PAIR_CONV and stride == 2 and K == 27 and Cout % 4 == 0 and Cin % 32 == 0 and \
                Cin >= PAIR_CONV_MIN_CIN and _nearly_empty_map(in_cs, out_cs):
            # pair-list kernel in exact fp32 (the stem of the 256-channel configuration)
            pair_cap = (min(27 * out_cs.n, 8 * in_cs.n) + 128 * K + 127) // 128 * 128
            pw_bytes = _lib.load().cnrma_sparse_conv_pairs_workspace_bytes(out_cs.n, K, Cout, pair_cap)
            pw = _workspace(pw_bytes, x.device)
            call("cnrma_sparse_conv_pairs_f32", ptr(x.F.contiguous()), Cin, ptr(nbr), K, ptr(w), Cout, ptr(scale), ptr(shift),
                 ptr(res), ACT[act], ptr(out), out_cs.n, ptr(out_cs.n_dev), pair_cap, ptr(pw), pw_bytes, stream())
            if P.static():                          # as in the f16x3 branch: a dropped pair entry must never be silent
                P.current().watch(pw[:PAIR_HDR_BYTES].view(torch.int32)[PAIR_OVERFLOW_WORD:PAIR_OVERFLOW_WORD + 1].clone(), 0, 0)
        elif prec == "f32" and GO_F32 and K == 27 and stride == 1 and Cout >= 64 and Cin % 32 == 0 and _gather_once(in_cs, out_cs):
            # exact fp32 on the gather-once structure (v_mfma_f32_32x32x2_f32 over the tile unions)
            go_ws_bytes = out_cs.n * Cout * 4 * (Cin // 32) if out_cs.n < GO_WS_ROWS else 0
            go_ws = _workspace(go_ws_bytes, x.device) if go_ws_bytes else None
            call("cnrma_sparse_conv_go_f32", ptr(x.F.contiguous()), Cin, ptr(tile_union(in_cs, out_cs, kernel_size, in_cs.stride)),
                 ptr(weights_f32_frag(weight)), Cout, ptr(scale), ptr(shift), ptr(res), ACT[act], ptr(out), out_cs.n,
                 ptr(out_cs.n_dev), ptr(go_ws), go_ws_bytes, stream())
        else:
            call("cnrma_sparse_conv_f32", ptr(x.F.contiguous()), Cin, ptr(nbr), K, ptr(w), Cout, ptr(scale), ptr(shift),
                 ptr(res), ACT[act], ptr(out), out_cs.n, ptr(out_cs.n_dev), ptr(ws), ws_bytes, stream())
    return SparseTensor(out, out_cs, out_split)


TRAIN_GO = "auto"       # bf16 training: forward and data gradient of the same-coordinates 3x3x3 convolutions on the gather-once
                        # kernel (cnrma_sparse_conv_go_bf16) where the inference path would use it; True / False force it


def _train_go(sets, n_out, Cin, Cout):
    """(tile unions, True) when a bf16 convolution of the training step with Cin -> Cout runs on the gather-once kernel"""
    if sets is None or TRAIN_GO is False or Cin % 32 or Cout < 64 or n_out == 0:
        return False
    in_cs, out_cs, ksize = sets
    if out_cs is not in_cs or ksize != 3:
        return False
    return True if TRAIN_GO is True else _gather_once(in_cs, out_cs)


def _conv_go_bf16(feats, sets, image, Cin, Cout):
    in_cs, out_cs, ksize = sets
    n = out_cs.n
    out = torch.empty((n, Cout), dtype=torch.float32, device=feats.device)
    go_ws_bytes = n * Cout * 4 * (Cin // 32) if n < GO_WS_ROWS else 0
    go_ws = _workspace(go_ws_bytes, feats.device) if go_ws_bytes else None
    call("cnrma_sparse_conv_go_bf16", ptr(feats.contiguous().float()), Cin, ptr(tile_union(in_cs, out_cs, ksize, in_cs.stride)),
         ptr(image), Cout, ptr(out), n, None, ptr(go_ws), go_ws_bytes, stream())
    return out


def _dgrad_weights(w3, flip, precision):
    """the weights of the data gradient, W'[k] = W[K - 1 - k if flip else k]^T: under bf16 the prepared image straight
    from W (one kernel; the image cannot be cached -- the optimiser changes W every step), else the fp32 tensor"""
    K, Cin, Cout = w3.shape
    if precision == "bf16" and Cout % 32 == 0:
        img = torch.empty(_lib.load().cnrma_sparse_conv_bf16_weight_bytes(K, Cout, Cin), dtype=torch.uint8, device=w3.device)
        call("cnrma_sparse_conv_prepare_weights_bf16_t", ptr(w3.contiguous()), K, Cin, Cout, 1 if flip else 0, ptr(img), stream())
        return (K, Cout, Cin), img
    wt = (w3.flip(0) if flip else w3).transpose(1, 2).contiguous()
    return (K, Cout, Cin), wt


@torch.no_grad()
def _conv_on_table(feats, n_out, nbr, weight, precision=None, bf16_image=None):
    """out [n_out, Cout] = sum_k feats[nbr[:, k]] @ weight[k] for an explicit neighbour table (None = identity, K == 1);
    bf16_image: ((K, Cin, Cout), prepared image) instead of `weight` (precision "bf16")"""
    if bf16_image is not None:
        (K, Cin, Cout), image = bf16_image
        assert _precision(precision) == "bf16" and Cin % 32 == 0
        out = torch.empty((n_out, Cout), dtype=torch.float32, device=feats.device)
        if n_out == 0:
            return out
        ws_bytes = _lib.load().cnrma_sparse_conv_workspace_bytes(n_out, Cout, K)
        ws = _workspace(ws_bytes, feats.device) if ws_bytes else None
        call("cnrma_sparse_conv_bf16", ptr(feats.contiguous().float()), Cin, ptr(nbr), K, ptr(image), Cout,
             None, None, None, 0, ptr(out), n_out, None, ptr(ws), ws_bytes, stream())
        return out
    w = weight.detach().contiguous().float()
    if w.dim() == 2:
        w = w.unsqueeze(0)
    K, Cin, Cout = w.shape
    out = torch.empty((n_out, Cout), dtype=torch.float32, device=feats.device)
    if n_out == 0:
        return out
    ws_bytes = _lib.load().cnrma_sparse_conv_workspace_bytes(n_out, Cout, K)
    ws = _workspace(ws_bytes, feats.device) if ws_bytes else None
    feats = feats.contiguous().float()
    prec = _precision(precision)
    if prec == "bf16" and Cin % 32 == 0:
        call("cnrma_sparse_conv_bf16", ptr(feats), Cin, ptr(nbr), K, ptr(weights_bf16(weight)), Cout,
             None, None, None, 0, ptr(out), n_out, None, ptr(ws), ws_bytes, stream())
    elif prec == "f16x3" and Cin % 32 == 0:
        amax = torch.zeros(_AMAX_WORDS, dtype=torch.float32, device=feats.device)
        call("cnrma_absmax_f32", ptr(feats), feats.shape[0], None, Cin, ptr(amax), stream())
        call("cnrma_sparse_conv_f16x3", ptr(feats), ptr(amax), Cin, ptr(nbr), K, ptr(split_weights_f16(weight)), Cout, None, None,
             None, 0, ptr(out), None, n_out, None, ptr(ws), ws_bytes, stream())
    else:
        call("cnrma_sparse_conv_f32", ptr(feats), Cin, ptr(nbr), K, ptr(w), Cout, None, None, None, 0, ptr(out), n_out, None,
             ptr(ws), ws_bytes, stream())
    return out


WGRAD_BLOCKS = 1024     # blocks (of 8 waves) the weight-gradient launch aims for
WGRAD_GO = "auto"       # bf16 weight gradient on the tile unions (cnrma_sparse_conv_wgrad_go_bf16): "auto" / True / False
WGRAD_GO_BLOCKS = 256   # blocks that launch aims for (one per CU: 256 registers per lane)
WGRAD_GO_MIN_ROWS = 256
DGRAD_MIRROR = True     # data gradient of a same-coordinates convolution on the forward table (mirrored offsets)


def _conv_fwd(st, F, weight):
    """forward of the differentiable convolution; st: an object with nbr, n_out, precision, sets (an autograd ctx)"""
    # the Parameter object itself goes down (its prepared image is cached on it until the optimiser changes it)
    if st.precision == "bf16" and weight.dim() == 3 and weight.shape[0] == 27 and \
            _train_go(st.sets, st.n_out, weight.shape[1], weight.shape[2]):
        with torch.no_grad():
            return _conv_go_bf16(F.detach(), st.sets, weights_bf16_frag(weight), weight.shape[1], weight.shape[2])
    return _conv_on_table(F.detach().float(), st.n_out, st.nbr, weight, st.precision)


def _conv_bwd(st, F, weight, grad_out, need_F, need_W):
    """data and weight gradient of the convolution (st as in _conv_fwd, + symmetric)"""
    nbr, n_out = st.nbr, st.n_out
    w = weight.detach().float()
    w3 = w.unsqueeze(0) if w.dim() == 2 else w
    K, Cin, Cout = w3.shape
    g = grad_out.contiguous().float()
    n_in = F.shape[0]
    grad_F = grad_W = None
    if need_F:
        # same coordinates on both sides (odd kernel, stride 1): nbr_t[i][k] = nbr[i][K - 1 - k] -- the forward table
        # with the offsets of the weights mirrored; else the table is transposed by a kernel
        flip = bool(st.symmetric and nbr is not None and n_in == n_out and DGRAD_MIRROR)
        if nbr is None or flip:
            nbr_t = nbr
        else:
            nbr_t = torch.empty((n_in, K), dtype=torch.int32, device=g.device)
            if n_in and n_out:
                call("cnrma_sparse_kernel_map_transpose", ptr(nbr), n_out, None, K, n_in, ptr(nbr_t), stream())
            else:
                nbr_t.fill_(-1)
        if flip and st.precision == "bf16" and K == 27 and _train_go(st.sets, n_in, Cout, Cin):
            # the gather-once kernel on the forward's tile unions, weights mirrored + transposed in fragment order
            grad_F = _conv_go_bf16(g, st.sets, weights_bf16_frag(weight, transposed=True), Cout, Cin)
            shape_t = wt = None
        else:
            shape_t, wt = _dgrad_weights(w3, flip, st.precision)
        if wt is None:
            pass
        elif wt.dtype == torch.uint8:
            grad_F = _conv_on_table(g, n_in, nbr_t, None, st.precision, bf16_image=(shape_t, wt))
        else:
            grad_F = _conv_on_table(g, n_in, nbr_t, wt, st.precision)
    if need_W and _wgrad_go(st, K, Cin, Cout):
        # bf16, 27 offsets, compact rows: on the tile unions of the forward's gather-once structure (cached on the
        # coordinate set: the convolutions of a residual stage share them)
        in_cs, out_cs, ksize = st.sets
        tu = tile_union(in_cs, out_cs, ksize, in_cs.stride)
        per = 2 * ((Cin + 63) // 64) * ((Cout + 63) // 64)
        tiles = (n_out + 63) // 64
        # parts: a block takes ~2.9 us per tile of its part, a part costs its slab written and read back (~3 TB/s):
        # T(parts) = tiles / parts * 2.9 + parts * slab_us, at most one block per CU
        slab_us = 27 * Cin * Cout * 8 / 3e6
        parts = max(1, min(tiles, WGRAD_GO_BLOCKS // per, int(round((tiles * 2.9 / slab_us) ** 0.5))))
        slabs = torch.empty((parts, K, Cin, Cout), dtype=torch.float32, device=g.device)
        call("cnrma_sparse_conv_wgrad_go_bf16", ptr(F.detach().contiguous().float()), Cin, ptr(tu), ptr(g), Cout, n_out, None,
             parts, ptr(slabs), stream())
        grad_W = (slabs[0] if parts == 1 else slabs.sum(dim=0)).view(weight.shape)
    elif need_W:
        # one block of 8 waves per (row chunk, offset, 64x64 weight tile), one slab per block: chunks sized for ~1024
        # blocks (8 waves per SIMD over the launch), at least 8 steps of 32 rows each
        tiles = ((Cin + 63) // 64) * ((Cout + 63) // 64)
        want = max(1, WGRAD_BLOCKS // (K * tiles))
        rows = max(256, -(-max(n_out, 1) // want))
        rows = (rows + 31) // 32 * 32
        chunks = _lib.load().cnrma_sparse_conv_wgrad_chunks(max(n_out, 1), rows)
        # the kernel writes every element of every slab
        slabs = (torch.empty if n_out else torch.zeros)((chunks, K, Cin, Cout), dtype=torch.float32, device=g.device)
        if n_out:
            # under autocast(bf16) the weight gradient is a bf16 x bf16 -> fp32 reduction too (what AMP computes)
            call("cnrma_sparse_conv_wgrad_bf16" if st.precision == "bf16" else "cnrma_sparse_conv_wgrad_f32",
                 ptr(F.detach().contiguous().float()), Cin, ptr(nbr), K, ptr(g), Cout, n_out, None, rows, ptr(slabs), stream())
        grad_W = (slabs[0] if chunks == 1 else slabs.sum(dim=0)).view(weight.shape)
    return grad_F, grad_W


class _ConvFn(torch.autograd.Function):
    """sum_k F[nbr[:, k]] @ W[k] with gradients: dgrad = the same convolution of grad_out over the transposed table with
    W[k]^T (or the forward table with mirrored offsets), wgrad = cnrma_sparse_conv_wgrad_* (SURVEY.md 8f rank 3)."""

    @staticmethod
    def forward(ctx, F, weight, nbr, n_out, precision, symmetric=False, sets=None):
        ctx.save_for_backward(F, weight)
        precision = _precision(precision)          # resolved here: the backward runs outside the autocast region
        ctx.nbr, ctx.n_out, ctx.precision, ctx.symmetric, ctx.sets = nbr, n_out, precision, symmetric, sets
        return _conv_fwd(ctx, F, weight)

    @staticmethod
    def backward(ctx, grad_out):
        F, weight = ctx.saved_tensors
        grad_F, grad_W = _conv_bwd(ctx, F, weight, grad_out, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return grad_F, grad_W, None, None, None, None, None


class _ConvBnActFn(torch.autograd.Function):
    """convolution -> BatchNorm (training) -> [+ residual] -> [ReLU / ELU] as ONE autograd node (the training forward is bound by
    the host's launch rate: one Function.apply, no intermediate SparseTensor / module calls per layer).  The same kernels
    as _ConvFn followed by _BatchNormTrainFn."""

    @staticmethod
    def forward(ctx, F, weight, bn_w, bn_b, residual, nbr, n_out, precision, symmetric, sets, eps, act, momentum,
                running_mean, running_var, batches):
        ctx.nbr, ctx.n_out, ctx.precision, ctx.symmetric, ctx.sets = nbr, n_out, _precision(precision), symmetric, sets
        y0 = _conv_fwd(ctx, F, weight)                           # the BatchNorm's input
        n, C = y0.shape
        out = torch.empty_like(y0)
        ws = torch.empty(_lib.load().cnrma_instnorm_workspace_bytes(C) // 8, dtype=torch.float64, device=y0.device)
        w = bn_w.detach().contiguous().view(-1).float()
        b = bn_b.detach().contiguous().view(-1).float()
        r = residual.detach().contiguous().float() if residual is not None else None
        call("cnrma_bn_train_forward_f32", ptr(y0), n, C, ptr(w), ptr(b), float(eps), ptr(r), act, float(momentum),
             ptr(running_mean), ptr(running_var), ptr(batches), ptr(out), ptr(ws), stream())
        ctx.save_for_backward(F, weight, y0, w, out if act else None)
        ctx.eps, ctx.ws, ctx.has_res, ctx.act = float(eps), ws, residual is not None, act
        return out

    @staticmethod
    def backward(ctx, grad_out):
        F, weight, y0, w, y = ctx.saved_tensors
        n, C = y0.shape
        g = grad_out.contiguous().float()
        dx = torch.empty_like(y0)
        dres = torch.empty_like(y0) if ctx.has_res and y is not None else None
        dw = torch.empty(C, dtype=torch.float32, device=y0.device)
        db = torch.empty(C, dtype=torch.float32, device=y0.device)
        ws2 = torch.empty(_lib.load().cnrma_instnorm_workspace_bytes(C) // 8, dtype=torch.float64, device=y0.device)
        call("cnrma_bn_train_backward_f32", ptr(g), ptr(y0), ptr(y), ctx.act, n, C, ptr(ctx.ws), ptr(w), ctx.eps, ptr(dx), ptr(dres),
             ptr(dw), ptr(db), ptr(ws2), stream())
        if ctx.has_res and dres is None:
            dres = g
        grad_F, grad_W = _conv_bwd(ctx, F, weight, dx, ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return (grad_F, grad_W, dw, db, dres) + (None,) * 11


FUSE_CONV_BN = True     # training: conv -> BatchNorm -> [+ shortcut] -> activation as one autograd node where the fused BatchNorm applies


def conv_bn_act_train(x, weight, bn, kernel_size=3, stride=1, act=None, residual=None, precision=None):
    """training-mode conv -> bn -> [+ residual.F] -> act (None / "relu" / "elu") on SparseTensor x; one autograd node when
    the fused BatchNorm takes the shape (see batch_norm_train), the composition of conv_autograd and batch_norm_train else"""
    _lib.require_gpu()
    Cout = weight.shape[-1]
    res_f = None if residual is None else residual.F
    if (not FUSE_CONV_BN or not BN_TRAIN_HIP or Cout > 256 or Cout % 4 or not bn.affine or not bn.track_running_stats
            or bn.momentum is None or (res_f is not None and res_f.dtype != torch.float32)):
        y = conv_autograd(x, weight, kernel_size, stride, precision)
        return SparseTensor(batch_norm_train(y.F, bn, act if act else False, res_f), y.cs)
    K = kernel_size ** 3
    in_cs = x.cs
    out_cs = in_cs if stride == 1 else in_cs.strided(stride)
    if out_cs.n < 2:
        y = conv_autograd(x, weight, kernel_size, stride, precision)
        return SparseTensor(batch_norm_train(y.F, bn, act if act else False, res_f), y.cs)
    nbr = None if (kernel_size == 1 and stride == 1) else in_cs.neighbours(out_cs, kernel_size, in_cs.stride)
    assert (weight.shape[0] if weight.dim() == 3 else 1) == K
    out = _ConvBnActFn.apply(x.F, weight, bn.weight, bn.bias, res_f, nbr, out_cs.n, precision,
                             out_cs is in_cs and kernel_size % 2 == 1, (in_cs, out_cs, kernel_size) if K == 27 else None,
                             bn.eps, {None: 0, "relu": 1, "elu": 2}[act], bn.momentum, bn.running_mean, bn.running_var,
                             bn.num_batches_tracked)
    return SparseTensor(out, out_cs)


def _wgrad_go(ctx, K, Cin, Cout):
    """weight gradient on the gather-once structure: bf16 mode, a 27-offset table between known coordinate sets whose rows
    are compact (WGRAD_GO "auto") -- a tile's offsets then share most of their input rows; True / False force the choice"""
    if ctx.precision != "bf16" or ctx.sets is None or K != 27 or ctx.n_out == 0 or Cin % 4 or Cout % 4 or WGRAD_GO is False:
        return False
    if WGRAD_GO is True:
        return True
    in_cs, out_cs, _ = ctx.sets
    # a coarsening convolution whose outputs merge several inputs each reads ~27 DISTINCT rows per output: the unions of its
    # tiles do not fit the image (many offset groups, staged in place) -- the block kernel is the faster one there
    # (measured at S: 64 -> 128 over 72.9 k -> 11.2 k rows 150 us against 49).  A stride over a sparse sample merges nothing.
    if out_cs is not in_cs and in_cs.n > 1.6 * ctx.n_out:
        return False
    return bool(in_cs.compact and ctx.n_out >= WGRAD_GO_MIN_ROWS)


def conv_autograd(x, weight, kernel_size=3, stride=1, precision=None):
    """differentiable MinkowskiConvolution (no fused epilogue): gradients flow to x.F and to the weight"""
    _lib.require_gpu()
    K = kernel_size ** 3
    in_cs = x.cs
    out_cs = in_cs if stride == 1 else in_cs.strided(stride)
    nbr = None if (kernel_size == 1 and stride == 1) else in_cs.neighbours(out_cs, kernel_size, in_cs.stride)
    assert (weight.shape[0] if weight.dim() == 3 else 1) == K
    out = _ConvFn.apply(x.F, weight, nbr, out_cs.n, precision, out_cs is in_cs and kernel_size % 2 == 1,
                        (in_cs, out_cs, kernel_size) if K == 27 else None)
    return SparseTensor(out, out_cs)


def _train(*tensors):
    """training path wanted: autograd is on and one of the operands carries a gradient"""
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


def _act_torch(f, act):
    return torch.relu(f) if act == "relu" else (torch.nn.functional.elu(f) if act == "elu" else f)


def conv_transpose_generative(x, weight, scale=None, shift=None, act=None, precision=None):
    """MinkowskiGenerativeConvolutionTranspose(k=2, s=2): 8 children per parent at half the tensor stride;
    out row 8*i + m = in[i] @ W[k]: parent-major, the children in Morton order (m = x<<2 | y<<1 | z for the child offset
    bits; the weight slice k decodes with x fastest as in the reference's kernel offsets)."""
    _lib.require_gpu()
    if _train(x.F, weight):
        # training: 8 dense GEMMs (no neighbour structure: every parent has all 8 children) through torch / rocBLAS
        n, half = x.cs.n, x.cs.stride // 2
        m = torch.arange(8, device=x.device)
        off = torch.stack((torch.zeros_like(m), (m >> 2) & 1, (m >> 1) & 1, m & 1), dim=1).to(torch.int32) * half
        k_of_m = ((m >> 2) & 1) | (m & 2) | ((m & 1) << 2)
        out_c = (x.C.unsqueeze(1) + off.unsqueeze(0)).reshape(8 * n, 4).contiguous()
        out_f = torch.einsum("nc,kcd->nkd", x.F, weight.float()[k_of_m]).reshape(8 * n, weight.shape[2]).float()   # fp32 storage under autocast too
        if scale is not None:
            out_f = out_f * scale
        if shift is not None:
            out_f = out_f + shift
        cs = CoordSet(out_c, half, None, x.cs.n_batch)
        cs.compact = x.cs.compact
        cs._gen_parent = x.cs
        return SparseTensor(_act_torch(out_f, act), cs)
    w = weight.contiguous().float()
    K, Cin, Cout = w.shape
    assert K == 8 and x.cs.stride % 2 == 0
    n = x.cs.n
    half = x.cs.stride // 2
    out_c = torch.empty((8 * n, 4), dtype=torch.int32, device=x.device)
    out_f = torch.empty((8 * n, Cout), dtype=torch.float32, device=x.device)
    out_split = None
    out_amax = None
    if n:
        prec = precision or CONV_PRECISION
        if P.static():
            P.current().keep(weight, w, scale, shift)
        if prec == "f16x3" and Cin % 32 == 0:
            out_amax = _amax_slot(x.device)
            call("cnrma_sparse_convtr_gen_f16x3", ptr(x.C), ptr(x.F.contiguous()), ptr(x.absmax()), n, ptr(x.cs.n_dev), Cin, half,
                 ptr(split_weights_f16(weight)), Cout, ptr(scale), ptr(shift), ACT[act], ptr(out_c), ptr(out_f),
                 ptr(out_amax), stream())
        elif prec == "bf16x6" and Cin % 32 == 0:
            if PRESPLIT and Cout % 8 == 0:
                out_split = torch.empty((8 * n + 1, Cout // 8, 3, 8), dtype=torch.bfloat16, device=x.device)
            in_split = x.split() if (PRESPLIT or x._split is not None) else None
            call("cnrma_sparse_convtr_gen_bf16x6", ptr(x.C), ptr(x.F.contiguous()), ptr(in_split), n, ptr(x.cs.n_dev), Cin, half,
                 ptr(split_weights(weight)), Cout, ptr(scale), ptr(shift), ACT[act], ptr(out_c), ptr(out_f),
                 ptr(out_split), stream())
        else:
            call("cnrma_sparse_convtr_gen_f32", ptr(x.C), ptr(x.F.contiguous()), n, ptr(x.cs.n_dev), Cin, half, ptr(w), Cout,
                 ptr(scale), ptr(shift), ACT[act], ptr(out_c), ptr(out_f), stream())
    # the kernels write child m of parent i at row 8 * i + m: the live rows stay contiguous, and 64 consecutive rows are the
    # children of 8 consecutive parents (compact tiles for the gather-once convolution when the parents' rows are)
    nd = x.cs.n_dev * 8 if x.cs.n_dev is not None else None
    cs = CoordSet(out_c, half, None, x.cs.n_batch, n_dev=nd)
    cs.compact = x.cs.compact
    cs.sorted = x.cs.sorted           # child key = parent key with the child's rank in the three bits below it
    cs._gen_parent = x.cs
    return SparseTensor(out_f, cs, out_split, out_amax)


class _MaxPoolFn(torch.autograd.Function):
    """max over the window through the pooling kernel; backward: every input row has exactly ONE parent (k = stride), so
    grad_in[i] = grad_out[parent(i)] where F[i] attains the parent's maximum -- plain gathers, no scatter"""

    @staticmethod
    def forward(ctx, F, nbr, n_out):
        Fd = F.detach().contiguous()
        C = Fd.shape[1]
        out = torch.empty((n_out, C), dtype=torch.float32, device=Fd.device)
        if n_out:
            call("cnrma_sparse_maxpool_f32", ptr(Fd), C, ptr(nbr), nbr.shape[1], ptr(out), n_out, None, stream())
        ctx.save_for_backward(Fd, out)
        ctx.nbr = nbr
        return out

    @staticmethod
    def backward(ctx, grad_out):
        Fd, out = ctx.saved_tensors
        nbr = ctx.nbr
        n_in, K = Fd.shape[0], nbr.shape[1]
        nbr_t = torch.empty((n_in, K), dtype=torch.int32, device=Fd.device)
        call("cnrma_sparse_kernel_map_transpose", ptr(nbr), out.shape[0], None, K, n_in, ptr(nbr_t), stream())
        parent = nbr_t.max(dim=1).values.long()
        has = parent >= 0
        parent = parent.clamp(min=0)
        hit = (Fd == out.index_select(0, parent)) & has.unsqueeze(1)
        return grad_out.contiguous().index_select(0, parent) * hit.to(grad_out.dtype), None, None


def max_pool(x, kernel_size=2, stride=2):
    _lib.require_gpu()
    out_cs = x.cs.strided(stride)
    nbr = x.cs.neighbours(out_cs, kernel_size, x.cs.stride)
    C = x.F.shape[1]
    if _train(x.F):      # training: the kernel's maximum; the gradient goes to the child that attains it
        return SparseTensor(_MaxPoolFn.apply(x.F, nbr, out_cs.n), out_cs)
    out = torch.empty((out_cs.n, C), dtype=torch.float32, device=x.device)
    if out_cs.n:
        call("cnrma_sparse_maxpool_f32", ptr(x.F.contiguous()), C, ptr(nbr), nbr.shape[1], ptr(out), out_cs.n,
             ptr(out_cs.n_dev), stream())
    return SparseTensor(out, out_cs, None, x.amax)       # a maximum over a subset: the input's bound still holds


def instance_norm_max_pool(x, weight=None, bias=None, eps=1e-8, relu=True, kernel_size=2, stride=2):
    """max_pool(instance_norm(x, relu=relu)) of a single-scene tensor without the normalised intermediate (the stem of the
    backbone): one statistics pass + one pooling pass that normalises its candidates on the fly; bit-identical to the two
    separate operators, and the pooled tensor carries its magnitude bound."""
    _lib.require_gpu()
    n, C = x.F.shape
    assert x.cs.n_batch <= 1 and C % 4 == 0 and not _train(x.F, weight, bias)
    ws = torch.empty(_lib.load().cnrma_instnorm_workspace_bytes(C) // 8, dtype=torch.float64, device=x.device)
    w = weight.contiguous().view(-1).float() if weight is not None else None
    b = bias.contiguous().view(-1).float() if bias is not None else None
    src = x.F.contiguous()
    if P.static():
        P.current().keep(w, b)
    out_cs = x.cs.strided(stride)
    nbr = x.cs.neighbours(out_cs, kernel_size, x.cs.stride)
    out = torch.empty((out_cs.n, C), dtype=torch.float32, device=x.device)
    amax = _amax_slot(x.device)
    if n and out_cs.n:
        call("cnrma_sparse_instnorm_f32", ptr(src), n, ptr(x.cs.n_dev), None, C, ptr(w), ptr(b), float(eps), int(relu), None,
             ptr(ws), stream())
        call("cnrma_sparse_instnorm_maxpool_f32", ptr(src), C, ptr(ws), ptr(w), ptr(b), float(eps), int(relu), ptr(nbr),
             nbr.shape[1], ptr(out), out_cs.n, ptr(out_cs.n_dev), ptr(amax), stream())
    return SparseTensor(out, out_cs, None, amax)


def instance_norm(x, weight=None, bias=None, eps=1e-8, relu=False):
    """MinkowskiInstanceNorm for a single scene (+ optional fused ReLU)."""
    _lib.require_gpu()
    n, C = x.F.shape
    out = torch.empty_like(x.F)
    ws = torch.empty(_lib.load().cnrma_instnorm_workspace_bytes(C) // 8, dtype=torch.float64, device=x.device)
    w = weight.contiguous().view(-1).float() if weight is not None else None
    b = bias.contiguous().view(-1).float() if bias is not None else None
    if _train(x.F, weight, bias):    # training: per-scene statistics in torch
        outs, r0 = [], 0
        for nb in x.cs.batch_counts():
            f = x.F[r0:r0 + nb]
            mu = f.mean(dim=0, keepdim=True)
            var = ((f - mu) ** 2).mean(dim=0, keepdim=True)
            y = (f - mu) / torch.sqrt(var + eps)
            if weight is not None:
                y = y * weight.view(1, -1)
            if bias is not None:
                y = y + bias.view(1, -1)
            outs.append(torch.relu(y) if relu else y)
            r0 += nb
        return SparseTensor(torch.cat(outs), x.cs)
    src = x.F.contiguous()
    if P.static():
        P.current().keep(w, b)
    if x.cs.n_batch <= 1:
        if n:
            call("cnrma_sparse_instnorm_f32", ptr(src), n, ptr(x.cs.n_dev), None, C, ptr(w), ptr(b), float(eps), int(relu),
                 ptr(out), ptr(ws), stream())
    elif P.static():                        # per scene, the segment's offset / count read on the device
        assert x.cs.scene_major, "instance norm of a multi-scene tensor needs scene-major rows"
        cnt, off = x.cs.counts_dev()
        for b_ in range(x.cs.n_batch):
            call("cnrma_sparse_instnorm_f32", ptr(src), n, ptr(cnt[b_:b_ + 1]), ptr(off[b_:b_ + 1]), C, ptr(w), ptr(b),
                 float(eps), int(relu), ptr(out), ptr(ws), stream())
    else:                                   # statistics per scene (MinkowskiInstanceNorm): one pass per row segment
        assert x.cs.scene_major, "instance norm of a multi-scene tensor needs scene-major rows"
        r0 = 0
        for nb in x.cs.batch_counts():
            if nb:
                call("cnrma_sparse_instnorm_f32", ptr(src[r0:r0 + nb]), nb, None, None, C, ptr(w), ptr(b), float(eps),
                     int(relu), ptr(out[r0:r0 + nb]), ptr(ws), stream())
            r0 += nb
    return SparseTensor(out, x.cs)


BN_TRAIN_HIP = True      # training-mode BatchNorm on the library's kernels (deterministic fp64 column sums), fused with the ReLU /
                         # shortcut add + ReLU behind it, running statistics updated by the statistics kernel.  Rounds 3-4: off
                         # (its eight tiny running-statistics launches made the host-bound step slower: 34.8-35.5 vs 33.0-33.3 ms);
                         # end of round 5, fused: 24.1 vs 25.3 ms per step at the ScanNet shape on the same box


class _BatchNormTrainFn(torch.autograd.Function):
    """nn.BatchNorm1d in training mode over the rows of F [n, C] (MinkowskiBatchNorm), fused with what follows it in a residual
    block: out = [relu]( bn(F) [+ residual] ).  forward = cnrma_bn_train_forward_f32 (fp64 column sums in a fixed order; the
    statistics kernel also updates the running statistics and the batch counter), backward = cnrma_bn_train_backward_f32 (the
    ReLU's mask is taken from the saved output; the masked gradient is the residual branch's).  torch's own kernels for a tall
    [n, C] matrix (batch_norm_collect_statistics / _backward_reduce channels-last) read it at a fifth of HBM speed: 44 us of
    backward kernels per layer at the ScanNet shape against 30 here, plus the ReLU's own backward launch."""

    @staticmethod
    def forward(ctx, F, weight, bias, residual, eps, relu, momentum, running_mean, running_var, batches):
        Fd = F.detach().contiguous().float()
        n, C = Fd.shape
        out = torch.empty_like(Fd)
        ws = torch.empty(_lib.load().cnrma_instnorm_workspace_bytes(C) // 8, dtype=torch.float64, device=Fd.device)
        w = weight.detach().contiguous().view(-1).float()
        b = bias.detach().contiguous().view(-1).float()
        r = residual.detach().contiguous().float() if residual is not None else None
        act = {False: 0, None: 0, True: 1, "relu": 1, "elu": 2}[relu]
        call("cnrma_bn_train_forward_f32", ptr(Fd), n, C, ptr(w), ptr(b), float(eps), ptr(r), act, float(momentum),
             ptr(running_mean), ptr(running_var), ptr(batches), ptr(out), ptr(ws), stream())
        ctx.save_for_backward(Fd, w, out if act else None)
        ctx.eps, ctx.ws, ctx.has_res, ctx.act = float(eps), ws, residual is not None, act      # ws[:2C]: mean, biased variance (fp64)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        Fd, w, y = ctx.saved_tensors
        n, C = Fd.shape
        g = grad_out.contiguous().float()
        dx = torch.empty_like(Fd)
        dres = torch.empty_like(Fd) if ctx.has_res and y is not None else None
        dw = torch.empty(C, dtype=torch.float32, device=Fd.device)
        db = torch.empty(C, dtype=torch.float32, device=Fd.device)
        ws2 = torch.empty(_lib.load().cnrma_instnorm_workspace_bytes(C) // 8, dtype=torch.float64, device=Fd.device)
        call("cnrma_bn_train_backward_f32", ptr(g), ptr(Fd), ptr(y), ctx.act, n, C, ptr(ctx.ws), ptr(w), ctx.eps, ptr(dx), ptr(dres),
             ptr(dw), ptr(db), ptr(ws2), stream())
        if ctx.has_res and dres is None:
            dres = g                                    # no ReLU: the residual branch takes the incoming gradient as it is
        return dx, dw, db, dres, None, None, None, None, None, None


def batch_norm_train(F, bn, relu=False, residual=None):
    """training-mode forward of an nn.BatchNorm1d `bn` on F [n, C], optionally fused with a residual add and an activation
    (relu: False / True = "relu" / "elu"): act( bn(F) [+ residual] ), running statistics updated.  Through the HIP kernels where they take the shape (C <= 256,
    C % 4 == 0, affine, fixed momentum); torch otherwise"""
    n, C = F.shape
    if (not BN_TRAIN_HIP or not F.is_cuda or n < 2 or C > 256 or C % 4 or not bn.affine or not bn.track_running_stats
            or bn.momentum is None or F.dtype != torch.float32 or (residual is not None and residual.dtype != torch.float32)):
        out = bn(F)
        if residual is not None:
            out = out + residual
        return torch.nn.functional.elu(out) if relu == "elu" else (torch.relu(out) if relu else out)
    return _BatchNormTrainFn.apply(F, bn.weight, bn.bias, residual, bn.eps, relu, bn.momentum, bn.running_mean, bn.running_var,
                                   bn.num_batches_tracked)


def union_add(a, b):
    """`a + b` on different coordinate sets with equal tensor stride (fcaf3d_head.py:114)."""
    _lib.require_gpu()
    assert a.cs.stride == b.cs.stride and a.F.shape[1] == b.F.shape[1]
    if a.cs is b.cs:
        return SparseTensor(a.F + b.F, a.cs)
    # the result keeps the first operand's rows and appends the other's new ones: start from the generated children (the neck:
    # whole 8-blocks in Morton order), else from the larger set, so that the appended tail -- thin, without locality -- stays
    # short.  The reference's row order is a hash map's; the sum is commutative, bit for bit.  The choice must be the SAME in
    # the eager run and in the static trace of a scene (row order feeds the top-k tie-break): `cs.n` is a live count in one and
    # a capacity in the other, so the size comparison is a recorded branch of the plan.
    gen_a, gen_b = a.cs._gen_parent is not None, b.cs._gen_parent is not None
    if gen_a != gen_b:
        swap = gen_b
    else:
        plan = P.current()
        if plan is not None and plan.static:
            swap = bool(plan.next_flag())                    # None (calibration scenes disagreed) -> keep the operand order
        else:
            swap = b.cs.n > a.cs.n
            if plan is not None:
                plan.record_flag(swap)
    if swap:
        a, b = b, a
    na, nb, C = a.cs.n, b.cs.n, a.F.shape[1]
    dev = a.device
    m = CoordMap(na + nb, dev, arena=True)
    call("cnrma_sparse_build_map", ptr(a.C), na, ptr(a.cs.n_dev), ptr(m.keys), ptr(m.vals), m.cap, int(m.precleared), stream())
    out_c = torch.empty((na + nb, 4), dtype=torch.int32, device=dev)
    out_f = torch.empty((na + nb, C), dtype=torch.float32, device=dev)
    n_out = torch.empty(1, dtype=torch.int32, device=dev)
    ws = torch.empty(_lib.load().cnrma_union_workspace_bytes(nb), dtype=torch.uint8, device=dev)
    nbatch = max(a.cs.n_batch, b.cs.n_batch)
    out_cap = max(na, P.current().next_cap(na + nb)) if P.static() else na + nb
    call("cnrma_sparse_union_add_f32", ptr(a.C), ptr(a.F.contiguous()), na, ptr(a.cs.n_dev), ptr(b.C), ptr(b.F.contiguous()),
         nb, ptr(b.cs.n_dev), C, ptr(m.keys), ptr(m.vals), m.cap, ptr(out_c), ptr(out_f), out_cap, ptr(n_out), ptr(ws),
         stream())
    if P.static():
        P.current().watch(n_out, 0, out_cap)
        cs = CoordSet(out_c, a.cs.stride, m, nbatch, n=out_cap, n_dev=n_out)
        cs.compact = a.cs.compact and b.cs.compact        # a's rows, then b's new ones: two compact runs
        return SparseTensor(out_f[:out_cap], cs)
    if nbatch > 1:          # row count and rows per scene with ONE device->host read
        live = torch.arange(na + nb, device=dev) < n_out
        scene = out_c[:, 0]
        got = _lib.read_ints(torch.cat([n_out] + [((scene == s_) & live).sum().view(1).to(torch.int32)
                                                   for s_ in range(nbatch)]))
        n, counts = got[0], got[1:]
    else:
        n, counts = _count(n_out, na + nb)[0], None
    cs = CoordSet(out_c[:n], a.cs.stride, m, nbatch)
    cs._counts = counts
    cs.compact = a.cs.compact and b.cs.compact
    if _train(a.F, b.F):            # training: the kernel placed the rows (a's first, then b's new ones); sum through torch
        b_row = cs.neighbours(b.cs, 1, a.cs.stride, method="generic").view(-1).long()
        f = torch.zeros((n, C), dtype=torch.float32, device=dev)
        f = torch.cat((a.F, f[na:]))
        return SparseTensor(f.index_add(0, b_row, b.F), cs)
    return SparseTensor(out_f[:n], cs)


def interpolate(score, query_coords, n_dev=None):
    """score.features_at_coordinates(query): linear interpolation on score's lattice (fcaf3d_head.py:129).
    query_coords int32 [n,4]; n_dev: device word with the live number of queries (static trace)."""
    _lib.require_gpu()
    assert score.F.shape[1] == 1
    n = query_coords.shape[0]
    out = torch.empty((n, 1), dtype=torch.float32, device=score.device)
    m = score.cs.cmap
    if n:
        call("cnrma_sparse_interp_f32", ptr(query_coords.contiguous()), n, ptr(n_dev), ptr(score.F.contiguous()), ptr(m.keys),
             ptr(m.vals), m.cap, score.cs.stride, ptr(out), stream())
    return out


def prune(x, keep_mask, n_keep=None, counts=None):
    """MinkowskiPruning: rows where keep_mask, order preserved.  n_keep: number of ones in the mask when the caller
    knows it (saves the device->host read); counts: rows per scene of the result when known."""
    _lib.require_gpu()
    from .rma import mask_to_index
    n, C = x.F.shape
    mask = keep_mask.to(torch.uint8).contiguous()
    sel, n_sel = mask_to_index(mask)
    nd = None
    if P.static():            # n_keep = upper bound of the kept rows; their number stays on the device
        assert n_keep is not None
        k, nd = min(int(n_keep), n), n_sel
    else:
        k = _lib.read_ints(n_sel)[0] if n_keep is None else int(n_keep)
    out_c = torch.empty((k, 4), dtype=torch.int32, device=x.device)
    out_f = torch.empty((k, C), dtype=torch.float32, device=x.device)
    if n:
        call("cnrma_sparse_prune_f32", ptr(x.C), ptr(x.F.contiguous()), n, ptr(x.cs.n_dev), C, ptr(sel), ptr(out_c), ptr(out_f),
             stream())
    cs = CoordSet(out_c, x.cs.stride, None, x.cs.n_batch, n_dev=nd)
    cs._counts = counts
    cs.compact = x.cs.compact         # order preserved: a thinned-out compact set stays compact
    cs.sorted = x.cs.sorted
    if _train(x.F):
        return SparseTensor(x.F.index_select(0, torch.nonzero(mask).view(-1)), cs)
    return SparseTensor(out_f, cs, None, x.amax)


def head_post(y, coords, n_reg, n_cls, scale, voxel_size):
    """tail of FCAF3DHead.forward_single over the fused head GEMM output y [n, >= 1+R+n_cls] (one kernel)."""
    _lib.require_gpu()
    n, ldy = y.shape
    dev = y.device
    cen = torch.empty((n, 1), dtype=torch.float32, device=dev)
    box = torch.empty((n, n_reg), dtype=torch.float32, device=dev)
    cls = torch.empty((n, n_cls), dtype=torch.float32, device=dev)
    mx = torch.empty((n, 1), dtype=torch.float32, device=dev)
    pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
    if n:
        call("cnrma_fcaf3d_head_post_f32", ptr(y), ldy, ptr(coords), n, n_reg, n_cls, ptr(scale.detach().view(1)),
             float(voxel_size), ptr(cen), ptr(box), ptr(cls), ptr(mx), ptr(pts), stream())
    return cen, box, cls, mx, pts


def max_scores(cls_score, centerness):
    n, nc = cls_score.shape
    mx = torch.empty(n, dtype=torch.float32, device=cls_score.device)
    if n:
        call("cnrma_fcaf3d_max_score_f32", ptr(cls_score.contiguous()), ptr(centerness.contiguous()), n, nc, ptr(mx), stream())
    return mx


def select_decode(ids, cls_score, centerness, bbox_pred, points, yaw_parametrization="fcaf3d"):
    """scores (sigmoid(cls)*sigmoid(ctr)) and decoded boxes of the rows `ids` (None = all rows) in one kernel."""
    n, R = bbox_pred.shape
    nc = cls_score.shape[1]
    k = n if ids is None else ids.shape[0]
    if R == 6:
        mode, W = 0, 6
    elif yaw_parametrization == "naive":
        mode, W = 3, 7
    elif yaw_parametrization == "sin-cos":
        mode, W = 2, 7
    else:
        mode, W = 1, 7
    scores = torch.empty((k, nc), dtype=torch.float32, device=bbox_pred.device)
    boxes = torch.empty((k, W), dtype=torch.float32, device=bbox_pred.device)
    if k:
        call("cnrma_fcaf3d_select_decode_f32", ptr(ids.contiguous()) if ids is not None else None, k,
             ptr(cls_score.contiguous()), ptr(centerness.contiguous()), ptr(bbox_pred.contiguous()),
             ptr(points.contiguous().float()), nc, R, mode, ptr(scores), ptr(boxes), stream())
    return boxes, scores


def topk_mask(scores, k, n_dev=None):
    """uint8 keep-mask of the k largest scores (ties -> smaller index): row set of torch.topk(scores, k), no sort.
    n_dev: device word with the live number of scores (rows behind it are never kept)."""
    _lib.require_gpu()
    scores = scores.contiguous().view(-1).float()
    n = scores.numel()
    mask = torch.empty(n, dtype=torch.uint8, device=scores.device)
    if n_dev is None:
        n_dev = _n_word(scores.device, n)
    ws = torch.empty(_lib.load().cnrma_sample_workspace_bytes(), dtype=torch.uint8, device=scores.device)
    call("cnrma_topk_mask_f32", ptr(scores), ptr(n_dev), n, int(k), ptr(mask), ptr(ws), stream())
    return mask


_NDEV = {}


def _n_word(device, n):
    """device word holding the constant n (cached)"""
    key = (device, n)
    w = _NDEV.get(key)
    if w is None:
        if len(_NDEV) > 256:
            _NDEV.clear()
        w = _NDEV[key] = torch.full((1,), n, dtype=torch.int32, device=device)
    return w


def topk_indices(scores, k, n_dev=None):
    """int64 [k] row indices of the k largest scores in descending score order (ties -> smaller index): what
    torch.topk(scores, k)[1] returns, built from the radix-select keep-mask + a compaction + a sort of only the k survivors
    (torch.topk's single-workgroup select takes 0.2-0.7 ms on ~500 k scores and sits on the critical path of a scene).
    With fewer than k live rows (n_dev) the live rows come first, in score order; the remaining slots repeat row 0."""
    scores = scores.contiguous().view(-1).float()
    n = scores.numel()
    if 0 < k <= 1024 and n > 0:          # one select + one single-workgroup sort of the k survivors (2 + 8 launches, no torch ops)
        _lib.require_gpu()
        out = torch.empty(k, dtype=torch.int64, device=scores.device)
        ws = torch.empty(_lib.load().cnrma_sample_workspace_bytes(), dtype=torch.uint8, device=scores.device)
        call("cnrma_topk_indices_f32", ptr(scores), ptr(n_dev if n_dev is not None else _n_word(scores.device, n)), n, int(k),
             ptr(out), ptr(ws), stream())
        return out
    mask = topk_mask(scores, k, n_dev)
    sel = torch.empty(n, dtype=torch.int32, device=scores.device)
    n_sel = torch.empty(1, dtype=torch.int32, device=scores.device)
    ws = torch.empty(_lib.load().cnrma_scan_workspace_bytes(n), dtype=torch.uint8, device=scores.device)
    call("cnrma_mask_to_index", ptr(mask), ptr(sel), ptr(n_sel), n, ptr(ws), stream())     # sel[row] = output slot or -1
    rows = torch.zeros(k + 1, dtype=torch.int64, device=scores.device)                      # slot -> row (slot k: dump)
    rows.scatter_(0, torch.where(sel >= 0, sel, torch.full_like(sel, k)).long().clamp_(max=k),
                  torch.arange(n, device=scores.device, dtype=torch.int64))
    slot_ok = torch.arange(k, device=scores.device, dtype=torch.int32) < n_sel
    idx = torch.where(slot_ok, rows[:k], torch.zeros_like(rows[:k]))
    vals = torch.where(slot_ok, scores[idx], torch.full((k,), float("-inf"), device=scores.device))
    order = torch.sort(vals, descending=True, stable=True)[1]
    return idx[order]


def row_max(feats):
    n, C = feats.shape
    out = torch.empty((n, 1), dtype=torch.float32, device=feats.device)
    if n:
        call("cnrma_rowmax_f32", ptr(feats.contiguous()), n, None, C, ptr(out), stream())      # dead rows: harmless values
    return out


def decode_boxes(points_xyz, bbox_pred, yaw_parametrization="fcaf3d"):
    """FCAF3DHead._bbox_pred_to_bbox (fcaf3d_head.py:300-349)."""
    _lib.require_gpu()
    n, R = bbox_pred.shape
    if n == 0:
        return bbox_pred
    if R == 6:
        mode, W = 0, 6
    elif yaw_parametrization == "naive":
        mode, W = 3, 7
    elif yaw_parametrization == "sin-cos":
        mode, W = 2, 7
    else:
        mode, W = 1, 7
    out = torch.empty((n, W), dtype=torch.float32, device=bbox_pred.device)
    call("cnrma_fcaf3d_decode_f32", ptr(points_xyz.contiguous().float()), ptr(bbox_pred.contiguous().float()), R, n, mode,
         ptr(out), stream())
    return out


def class_scores(cls_score, centerness):
    """scores = sigmoid(cls) * sigmoid(centerness) and their per-row max (fcaf3d_head.py:249-250)."""
    _lib.require_gpu()
    n, nc = cls_score.shape
    scores = torch.empty((n, nc), dtype=torch.float32, device=cls_score.device)
    mx = torch.empty(n, dtype=torch.float32, device=cls_score.device)
    if n:
        call("cnrma_fcaf3d_scores_f32", ptr(cls_score.contiguous().float()), ptr(centerness.contiguous().float()), n, nc,
             ptr(scores), ptr(mx), stream())
    return scores, mx
