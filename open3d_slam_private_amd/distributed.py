"""Point-partitioned registration across the GPUs of one node.

The reading is split into contiguous slices, one per rank; the reference cloud and its voxel-bin
table are replicated.  Per Gauss-Newton iteration every rank runs R3-R7 on its slice and the ranks
exchange
  * (TrimmedDist only) the exact global `ratio`-quantile of the squared match distances by a
    3-level radix select: three all-reduces of a 2048-bin histogram (bits 31:21, 20:10, 9:0), and
  * one all-reduce (sum) of 32 doubles {21 upper-triangular H, 6 b, error, #inliers, #matched, sum d2}.
Every rank then solves the same 6x6 system redundantly (no broadcast).  `torch.distributed` is the
transport (backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

The reference has no distributed path at all (single process); this module's
contract is "same numbers as the single-rank path on the concatenated reading".
"""
from __future__ import annotations

import math
import time

import numpy as np


def select_from_hist(hist: np.ndarray, rank: int):
    """Bin containing the element of 0-based `rank`, and the rank inside that bin."""
    cum = np.cumsum(hist.astype(np.int64))
    b = int(np.searchsorted(cum, rank, side="right"))
    before = int(cum[b - 1]) if b > 0 else 0
    return b, rank - before


def trim_rank(n_finite: int, ratio: float) -> int:
    """Index used by Matches::getDistsQuantile (Matches.cpp:82-86): size()*quantile in float, truncated;
    quantile == 1 -> the maximum."""
    if np.float32(ratio) == np.float32(1.0):
        return max(n_finite - 1, 0)
    r = int(np.float32(n_finite) * np.float32(ratio))
    return min(r, max(n_finite - 1, 0))


class DistributedRegistration:
    """Drives one registration over all ranks of `group`.

    `local` is this rank's slice backend and must provide
        prepare_with_centroid / match_local(T) / trim_histogram(level, prefix) / reduce_local(T, limit)
    (capi.Registration on the GPU box; an oracle-backed stand-in in the gloo CPU tests).
    `solve_update(sums, T_iter) -> T_next` is the host-side R8/R9 step.
    """

    def __init__(self, local, solve_update, use_trimmed: bool, trim_ratio: float, fixed_iters: int = 20,
                 dist=None, group=None, device=None):
        self.local = local
        self.solve_update = solve_update
        self.use_trimmed = use_trimmed
        self.trim_ratio = trim_ratio
        self.fixed_iters = fixed_iters
        self.dist = dist
        self.group = group
        self.device = device

    def _allreduce(self, arr: np.ndarray) -> np.ndarray:
        if self.dist is None or self.dist.get_world_size(self.group) == 1:
            return arr
        import torch
        t = torch.from_numpy(np.ascontiguousarray(arr))
        if self.device is not None:
            t = t.to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy()

    def global_trim_limit(self) -> float:
        h0 = self._allreduce(self.local.trim_histogram(0, 0).astype(np.int64))
        n_finite = int(h0.sum())
        if n_finite == 0:
            return math.inf
        rank = trim_rank(n_finite, self.trim_ratio)
        b0, rank = select_from_hist(h0, rank)
        prefix = b0 << 21
        h1 = self._allreduce(self.local.trim_histogram(1, prefix).astype(np.int64))
        b1, rank = select_from_hist(h1, rank)
        prefix |= b1 << 10
        h2 = self._allreduce(self.local.trim_histogram(2, prefix).astype(np.int64))
        b2, rank = select_from_hist(h2[:1024], rank)
        prefix |= b2
        return float(np.array([prefix], np.uint32).view(np.float32)[0])

    def iterate(self, T_iter: np.ndarray):
        """One Gauss-Newton iteration over all ranks.  Returns (T_next, global sums)."""
        self.local.match_local(T_iter)
        limit = self.global_trim_limit() if self.use_trimmed else math.inf
        self.last_limit = limit
        sums = self._allreduce(self.local.reduce_local(T_iter, limit))
        return self.solve_update(sums, T_iter), sums

    def run(self, T_iter0=None):
        T = np.eye(4, dtype=np.float32) if T_iter0 is None else np.asarray(T_iter0, np.float32)
        sums = None
        for _ in range(self.fixed_iters):
            T, sums = self.iterate(T)
        return T, sums


class _DevArray:
    """Minimal __cuda_array_interface__ holder: lets torch alias device memory owned by the C library."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {"shape": shape, "typestr": typestr, "data": (int(ptr), False), "version": 2}


class StreamDistributedRegistration:
    """Same exchange as DistributedRegistration, but stream-ordered: per Gauss-Newton iteration the host only
    enqueues 5 kernel phases and (TrimmedDist) 3 + 1 all-reduces on ONE HIP stream -- no host synchronisation
    until the end.  `reg` is a capi.Registration whose stream is torch's current stream on `device`;
    `all_reduce(tensor)` defaults to torch.distributed.all_reduce (RCCL)."""

    def __init__(self, reg, use_trimmed, iters, dist=None, device=None, all_reduce=None):
        import torch
        self.reg, self.use_trimmed, self.iters, self.dist = reg, use_trimmed, iters, dist
        hist_ptr, sums_ptr = reg.dist_buffers()
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.hist = torch.as_tensor(_DevArray(hist_ptr, (3, 2048), "<i4"), device=dev)
        self.sums = torch.as_tensor(_DevArray(sums_ptr, (32,), "<f8"), device=dev)
        # R8x (use_xicp): two more small all-reduces, on the first iteration only
        self.use_xicp = bool(getattr(getattr(reg, "params", None), "use_xicp", 0))
        self._xicp_first = False
        if self.use_xicp:
            cp, sp = reg.dist_xicp_buffers()
            self.x_center = torch.as_tensor(_DevArray(cp, (4,), "<f8"), device=dev)
            self.x_sums = torch.as_tensor(_DevArray(sp, (12,), "<f8"), device=dev)
        if all_reduce is not None:
            self._ar = all_reduce
        elif dist is not None and dist.get_world_size() > 1:
            self._ar = lambda t: dist.all_reduce(t)
        else:
            self._ar = lambda t: None

    def _xicp_analysis(self):
        """First iteration with use_xicp: phase 4 only stashed the eigen-directions; collect the analysis sums over
        all ranks, then let every rank decide, solve and update identically (phase 9)."""
        reg = self.reg
        reg.dist_phase(7)
        self._ar(self.x_center)
        reg.dist_phase(8)
        self._ar(self.x_sums)
        reg.dist_phase(9)
        self._xicp_first = False

    def run(self, T_start=None):
        reg = self.reg
        reg.dist_begin(T_start)
        self._xicp_first = self.use_xicp
        for _ in range(self.iters):
            reg.dist_phase(0)
            if self.use_trimmed:
                self._ar(self.hist[0])
                reg.dist_phase(1)
                self._ar(self.hist[1])
                reg.dist_phase(2)
                self._ar(self.hist[2])
            reg.dist_phase(3)
            self._ar(self.sums)
            reg.dist_phase(4)
            if self._xicp_first:
                self._xicp_analysis()
        return reg.dist_finish()


class FusedStreamDistributedRegistration(StreamDistributedRegistration):
    """Stream-ordered multi-GPU loop that switches, once the trimmed limit has settled, from the select-based
    iteration (6 launches + 4 all-reduces) to the fused iteration (2 launches + ONE all-gather of a 66 KB block per
    rank).  The device decides whether a fused iteration was valid (exact verification of the predicted band, on
    every rank alike); a stalled iteration is repeated on the select-based path.  The decisions are taken by the
    library's own steering state machine (capi.Steer == reg_dist_steer_*, the code reg_dist_register runs in C++); this
    class only carries them out with torch.distributed collectives.  Kept for rehearsals with non-RCCL backends: the
    production N > 1 path is capi.Registration.dist_register."""

    def __init__(self, reg, use_trimmed, trim_ratio, iters, world, rank, dist=None, device=None, all_reduce=None,
                 all_gather=None, fixed=True, settle_tol=0.05, gather_select=True, n_max=None, timeout_s=30.0):
        super().__init__(reg, use_trimmed, iters, dist=dist, device=device, all_reduce=all_reduce)
        import torch
        self.world, self.rank, self.fixed, self.settle_tol, self.timeout_s = world, rank, fixed, settle_tol, timeout_s
        # settle_tol: a rank's contribution block holds a bounded number of band records (contrib_cap_for(world) in
        # kernels_fused.hpp), and the band of the first fused iterations is as wide as the limit still moves (measured: at
        # 25 % the first fused iteration overflowed the block, stalled, and the whole burst behind it ran as no-ops)
        # select-by-gather: ONE all-gather of the squared distances instead of three dependent histogram all-reduces
        # per select-based iteration (every rank then runs the exact select on the same multiset of values)
        self.gather_select = bool(gather_select) and bool(use_trimmed)
        if self.gather_select:
            if n_max is None:
                n_max = int(reg.n_source)
                if dist is not None and world > 1:
                    t = torch.tensor([n_max], dtype=torch.int64,
                                     device=device if device is not None else torch.device("cuda", torch.cuda.current_device()))
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    n_max = int(t.item())
            gdev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
            lp, ap = reg.dist_gather_buffers(world, n_max)
            self.d2_local = torch.as_tensor(_DevArray(lp, (n_max,), "<f4"), device=gdev)
            self.d2_all = torch.as_tensor(_DevArray(ap, (world * n_max,), "<f4"), device=gdev)
        self.trimming = bool(use_trimmed) and float(np.float32(trim_ratio)) != 1.0
        cp, gp, nbytes = reg.dist_fused_buffers(world, rank)
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        n = nbytes // 4
        self.contrib = torch.as_tensor(_DevArray(cp, (n,), "<f4"), device=dev)
        self.gathered = torch.as_tensor(_DevArray(gp, (world * n,), "<f4"), device=dev)
        if all_gather is not None:
            self._ag = all_gather
        elif dist is not None and world > 1:
            self._ag = lambda out, inp: dist.all_gather_into_tensor(out, inp)
        else:
            self._ag = lambda out, inp: out.copy_(inp)
        self.n_fused = self.n_generic = self.n_stalls = 0

    def _generic(self):
        reg = self.reg
        if self.gather_select:
            reg.dist_phase(10)
            self._ag(self.d2_all, self.d2_local)
            reg.dist_phase(11)
        else:
            reg.dist_phase(0)
            if self.use_trimmed:
                self._ar(self.hist[0])
                reg.dist_phase(1)
                self._ar(self.hist[1])
                reg.dist_phase(2)
                self._ar(self.hist[2])
            reg.dist_phase(3)
        self._ar(self.sums)
        reg.dist_phase(4)
        if self._xicp_first:
            self._xicp_analysis()

    def _fused(self):
        self.reg.dist_phase(5)
        self._ag(self.gathered, self.contrib)
        self.reg.dist_phase(6)

    def _deadline(self):
        return time.monotonic() + self.timeout_s

    def _record(self, seq_rel):
        """Blocks until sequence `seq_rel` has reported (returns its record) or can no longer report because the stream
        drained without it (returns None: an earlier sequence ended or stalled the loop).  Raises after `timeout_s`: a
        dead peer or a collective that never completes must not spin the survivors forever."""
        reg = self.reg
        t_end = self._deadline()
        while True:
            st = reg.dist_record(seq_rel)
            if int(st.sequences_done) == seq_rel:
                return st
            if st.stream_idle:
                st = reg.dist_record(seq_rel)          # the report may have landed between the two reads
                return st if int(st.sequences_done) == seq_rel else None
            if time.monotonic() > t_end:
                raise TimeoutError(f"sequence {seq_rel} of the distributed loop did not report within {self.timeout_s} s "
                                   "(a peer rank died or a collective never completed)")

    def _drain(self):
        reg = self.reg
        t_end = self._deadline()
        while not reg.dist_poll().stream_idle:
            if time.monotonic() > t_end:
                raise TimeoutError(f"the stream of the distributed loop did not drain within {self.timeout_s} s")
        return reg.dist_poll()

    def run(self, T_start=None):
        """Every decision (select-based or fused iteration, stop, repair) is taken by the library's steering state
        machine (reg_dist_steer_*, the SAME code reg_dist_register runs) from the record of ONE specific sequence,
        which is identical on every rank; nothing depends on how far the device happens to be when the host looks.
        Ranks therefore always enqueue the same collectives."""
        from . import capi
        reg = self.reg
        reg.dist_begin(T_start)
        self._xicp_first = self.use_xicp
        steer = capi.Steer(self.trimming, self.iters if self.fixed else 0, self.iters, self.settle_tol, True)
        reply = None
        while True:
            a = steer.step(reply)
            reply = None
            if a.kind == capi.STEER_DONE:
                break
            if a.kind == capi.STEER_RECORD:
                rec = self._record(int(a.seq))
                reply = capi.DistReply()
                if rec is not None:
                    reply.available, reply.iterations, reply.done, reply.stall = 1, int(rec.iterations), int(rec.done), int(rec.stall)
                    reply.limit_last, reply.limit_prev = float(rec.limit_last), float(rec.limit_prev)
            elif a.kind == capi.STEER_DRAIN:
                top = self._drain()
                reply = capi.DistReply()
                reply.available, reply.iterations, reply.done, reply.stall = 1, int(top.iterations), int(top.done), int(top.stall)
                reply.limit_last, reply.limit_prev = float(top.limit_last), float(top.limit_prev)
            elif a.kind == capi.STEER_GENERIC:
                self._generic()
            else:
                for _ in range(int(a.count)):
                    self._fused()
        self.n_generic, self.n_fused, self.n_stalls = steer.counts()
        return reg.dist_finish()


def host_staged_transport(dist, device, group=None):
    """(all_reduce_sum, all_gather) callables for capi.Registration.dist_init_custom that move the bytes with ANY
    torch.distributed backend through host staging (gloo on a box where RCCL cannot be used, e.g. several ranks sharing
    one GPU in a rehearsal).  Slow by construction (device -> host -> network -> device with full synchronisation); the
    production transport is RCCL inside the library (capi.Registration.dist_init)."""
    import torch
    from . import capi
    typ = {capi.DT_I32: "<i4", capi.DT_I64: "<i8", capi.DT_F64: "<f8"}
    world = dist.get_world_size(group)

    def all_reduce(buf, count, dtype, stream):
        try:
            torch.cuda.synchronize(device)
            t = torch.as_tensor(_DevArray(buf, (int(count),), typ[dtype]), device=device)
            c = t.cpu()
            dist.all_reduce(c, group=group)
            t.copy_(c)
            torch.cuda.synchronize(device)
            return 0
        except Exception:   # noqa: BLE001 -- reported to the library as a failed collective
            return 1

    def all_gather(send, recv, nbytes, stream):
        try:
            torch.cuda.synchronize(device)
            nbytes = int(nbytes)
            s_ = torch.as_tensor(_DevArray(send, (nbytes,), "|u1"), device=device).cpu()
            parts = [torch.empty_like(s_) for _ in range(world)]
            dist.all_gather(parts, s_, group=group)
            out = torch.as_tensor(_DevArray(recv, (world * nbytes,), "|u1"), device=device)
            out.copy_(torch.cat(parts))
            torch.cuda.synchronize(device)
            return 0
        except Exception:   # noqa: BLE001
            return 1

    return all_reduce, all_gather
