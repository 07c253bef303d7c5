"""Size plan of a scene forward: what lets the whole pass run without a single device->host read.

The reference sizes every intermediate tensor on the host (`nonzero()`, ME's coordinate manager: ray_marching.py:781,
:328-330) -- one synchronisation per data-dependent shape.  Here a forward is run ONCE eagerly under a recording Plan
(every data-dependent row count is read back as before and appended to `sizes`, every data-dependent branch to
`flags`); afterwards the same launch sequence runs in *static* mode: each such tensor gets a CAPACITY derived from the
recorded size (x margin), its live row count stays in a device word that the kernels read (the C-ABI takes a capacity and
a count pointer everywhere), and nothing is read back.  That sequence is what pipeline.StaticScene captures into a HIP
graph.  Every capacity / branch assumption is registered with `watch()`; `status()` folds them into one device word
(0 = all assumptions held) that travels with the detections, so a scene that outgrows its plan is detected and can be
re-run eagerly.
"""
import threading

import torch

_tls = threading.local()


def current():
    """the Plan active on this thread, or None (plain eager execution)"""
    return getattr(_tls, "plan", None)


def static():
    p = current()
    return p is not None and p.static


class using:
    def __init__(self, plan):
        self.plan = plan

    def __enter__(self):
        self.prev = current()
        _tls.plan = self.plan
        return self.plan

    def __exit__(self, *exc):
        _tls.plan = self.prev
        return False


def _round_up(n, q):
    return (n + q - 1) // q * q


ARENA = True      # static traces clear ONE 0xFF arena per scene instead of one table at a time (False: A/B, tests)


class Plan:
    def __init__(self, margin=1.2, slack=256):
        self.margin, self.slack = float(margin), int(slack)
        self.sizes, self.flags = [], []
        self.marks = {}            # name -> (len(sizes), len(flags)) at the time of mark(): splits the record sequence
        self.static = False
        self._reset()
        self._consts = []          # device constants of the static trace, created by the first (un-captured) static run
        self.workspaces = {}       # grow-only scratch buffers of the static trace (kept alive with the plan)
        self._counters = {}        # zeroed arrival counters (see counters())
        self._kept = {}            # id -> tensor: buffers the captured launches read through raw pointers but that the
                                   # modules may drop at any time (prepared weight images, folded BatchNorm vectors, fused head
                                   # weights: all cached per module and invalidated on train() / load_state_dict()) -- a graph
                                   # must never replay a pointer into freed allocator memory

    def _reset(self):
        self._i = self._j = self._c = 0
        self._watch = []
        self._amax = None
        # 0xFF arena of the static trace (hash tables, neighbour tables): the first static run of a plan sizes it (its tables
        # are allocated and cleared one by one), every later run -- the captured one -- takes its tables from ONE buffer that
        # is cleared by one launch at its first use
        if getattr(self, "static", False) and getattr(self, "_arena_need", 0) > 0 and getattr(self, "_arena", None) is None:
            self._arena = torch.empty(self._arena_need, dtype=torch.uint8, device=self._arena_dev)
        self._arena_off = 0
        self._arena_filled = False
        self.arena_fallbacks = 0   # tables of this trace that did not fit the arena sized by the plan's first static run

    # ---- recording (eager calibration run) --------------------------------------------------------------------------
    def record(self, n):
        assert not self.static
        self.sizes.append(int(n))
        return int(n)

    def mark(self, name):
        """calibration: remember where in the record sequence a stage begins (scaled() splits there)"""
        if not self.static:
            self.marks[name] = (len(self.sizes), len(self.flags))

    def scaled(self, n_scenes, split="net"):
        """the plan of a pass over `n_scenes` scenes derived from this single-scene plan: the records before the mark
        (the per-scene geometric half: one aggregation, one voxelisation per scene) are consumed once per scene, entry
        by entry; the records behind it (the network over the collated tensor) scale with the number of scenes"""
        i, j = self.marks[split]
        assert j == 0, "per-scene stages record sizes only"
        p = Plan(self.margin, self.slack)
        p.sizes = [s for s in self.sizes[:i] for _ in range(n_scenes)] + [s * n_scenes for s in self.sizes[i:]]
        p.flags = list(self.flags)
        p.marks = dict(self.marks)
        return p

    def record_flag(self, f):
        assert not self.static
        self.flags.append(bool(f))
        return bool(f)

    def merge(self, other):
        """fold another calibration run (another scene of the same configuration) into this plan: capacities cover the
        larger of the two sizes; a branch the two scenes took differently becomes None = "take the variant that is
        valid for both" (see next_flag callers)."""
        assert not self.static and len(self.sizes) == len(other.sizes) and len(self.flags) == len(other.flags), \
            "calibration runs of one configuration must record the same sequence"
        assert self.marks == other.marks
        self.sizes = [max(a, b) for a, b in zip(self.sizes, other.sizes)]
        self.flags = [a if a == b else None for a, b in zip(self.flags, other.flags)]
        return self

    # ---- static trace -------------------------------------------------------------------------------------------
    def begin_static(self):
        self.static = True
        self._reset()

    def end_static(self):
        assert self._i == len(self.sizes) and self._j == len(self.flags), \
            "the static trace consumed a different number of sizes / flags than the calibration run recorded"

    def next_cap(self, bound=None, n_dev=None, lo=0):
        """capacity of the next data-dependent tensor (recorded size x margin, clipped to a provable bound); n_dev = its
        live-count word: registered as `lo <= n <= capacity`"""
        n = self.sizes[self._i]
        self._i += 1
        cap = _round_up(int(n * self.margin) + self.slack, 64)
        if bound is not None:
            cap = min(cap, int(bound))
        cap = max(cap, 1)
        if n_dev is not None:
            self.watch(n_dev, lo, cap)
        return cap

    def next_flag(self):
        f = self.flags[self._j]
        self._j += 1
        return f

    def watch(self, n_dev, lo, hi):
        """assumption of the static trace: lo <= n_dev[0] <= hi"""
        self._watch.append((n_dev.view(-1)[:1], int(lo), int(hi)))

    def keep(self, *tensors):
        """pin tensors whose device pointers the static trace hands to kernels for as long as this plan (= its graph) lives"""
        for t in tensors:
            if t is not None:
                self._kept[id(t)] = t

    def const(self, builder):
        """a small device constant of the static trace: built by the first static run (outside graph capture, where a
        host->device copy is allowed), reused in sequence order by every later one"""
        if self._c == len(self._consts):
            self._consts.append(builder())
        t = self._consts[self._c]
        self._c += 1
        return t

    def workspace(self, nbytes, device):
        buf = self.workspaces.get(device)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self.workspaces[device] = buf
        return buf

    def arena_ff(self, nbytes, device):
        """static trace only: `nbytes` of device memory that hold 0xFF when the caller's kernel runs, or None when the caller has
        to allocate and clear by itself (calibration / eager runs; the first static run, which only sizes the arena)"""
        if not self.static or not ARENA:
            return None
        nbytes = (int(nbytes) + 255) // 256 * 256
        if getattr(self, "_arena", None) is None:
            self._arena_need = getattr(self, "_arena_need", 0) + nbytes
            self._arena_dev = device
            return None
        if self._arena_off + nbytes > self._arena.numel():          # a trace that asks for more than it did when it was sized:
            self.arena_fallbacks += 1                                # the caller clears its own table (correct, one launch more);
            return None                                              # counted, so that a drift between the sizing run and the
                                                                     # captured run shows (StaticScene info / test_static_gpu)
        if not self._arena_filled:
            from . import _lib
            _lib.call("cnrma_fill_bytes_u8", _lib.ptr(self._arena), 0xFF, self._arena.numel(), _lib.stream())
            self._arena_filled = True
        view = self._arena[self._arena_off:self._arena_off + nbytes]
        self._arena_off += nbytes
        return view

    def counters(self, device, words):
        """zeroed 32-bit words for kernels that count arrivals and leave them zero again (one buffer per plan: the launches
        of a scene are stream-ordered; scenes in flight have their own plans)"""
        buf = self._counters.get(device)
        if buf is None:
            buf = self._counters[device] = torch.zeros(words, dtype=torch.int32, device=device)
        return buf

    def amax_slot(self, device, words):
        """a zeroed magnitude bound for a conv epilogue, carved from chunks that are (re-)zeroed inside the trace"""
        if self._amax is None or self._amax[1] >= 64:
            self._amax = [torch.zeros(64 * words, dtype=torch.float32, device=device), 0]
        i = self._amax[1]
        self._amax[1] = i + 1
        return self._amax[0][i * words:(i + 1) * words]

    def status(self, device):
        """int32 [1] on the device: number of violated assumptions (0 = the static results are valid)"""
        if not self._watch:
            return torch.zeros(1, dtype=torch.int32, device=device)
        vals = torch.cat([w[0].to(torch.int32) for w in self._watch])
        lo = self.const(lambda: torch.tensor([w[1] for w in self._watch], dtype=torch.int32, device=device))
        hi = self.const(lambda: torch.tensor([w[2] for w in self._watch], dtype=torch.int32, device=device))
        if vals.numel() > 4096:
            return ((vals < lo) | (vals > hi)).sum().to(torch.int32).view(1)
        from . import _lib
        out = torch.empty(1, dtype=torch.int32, device=device)
        _lib.call("cnrma_range_violations_i32", _lib.ptr(vals), _lib.ptr(lo), _lib.ptr(hi), vals.numel(), _lib.ptr(out), _lib.stream())
        return out
