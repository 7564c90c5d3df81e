"""Utterance-level data parallelism for the AAS step: one process per GPU, RCCL over xGMI through
``torch.distributed`` (backend "nccl" on ROCm is RCCL; "gloo" for the CPU tests).

The reference has no distributed code; this is new functionality:
  * the length-sorted global minibatch is sharded by stride (rank r takes utterances r, r+W, ...),
    padding kept at the global max T so the un-masked L1 sums are exactly the single-process ones;
  * every loss is normalised by the GLOBAL normaliser (N_global, global nElement) before backward, so a
    plain SUM all-reduce of the flat E / D (and A, when trainable) gradient buffers is exact;
  * one small SUM all-reduce of the packed loss scalars gives every rank the same BEGAN kt;
  * A's BatchNorm uses local-batch statistics ("8 replicas with local-batch BN", documented).
No data-path collective other than these.
"""
import os

import torch
import torch.distributed as dist


_HOST_GROUPS = {}   # default-group backend -> a gloo group of all ranks for host-side control traffic (created collectively, once)


def _host_group():
    """A gloo group beside an `nccl` default group: host-side integers (the padded length of a sharded batch) and barriers
    that may outlast RCCL's watchdog (rank 0 validating for an hour) travel over it, without a device synchronisation and
    without a pending RCCL collective.  With a gloo default group (CPU tests) the default group itself is used."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    be = dist.get_backend()
    if be == "gloo":
        return None
    if be not in _HOST_GROUPS:
        import datetime
        # Single node (every rank of the job is local, or the rendezvous address is loopback): gloo's interface discovery
        # resolves the container hostname, which may not resolve here - pin the loopback interface for the CREATION of this
        # group only and put the caller's environment back.  Multi-node jobs keep gloo's own choice (or the user's
        # GLOO_SOCKET_IFNAME): loopback cannot rendezvous across nodes.
        world = dist.get_world_size()
        local = int(os.environ.get("LOCAL_WORLD_SIZE", "0") or 0)
        addr = os.environ.get("MASTER_ADDR", "")
        single_node = (local > 0 and local == world) or addr in ("127.0.0.1", "localhost", "::1")
        had, prev = "GLOO_SOCKET_IFNAME" in os.environ, os.environ.get("GLOO_SOCKET_IFNAME")
        if single_node and not had:
            os.environ["GLOO_SOCKET_IFNAME"] = "lo"
        try:
            _HOST_GROUPS[be] = dist.new_group(backend="gloo", timeout=datetime.timedelta(hours=24))
        finally:
            if single_node and not had:
                os.environ.pop("GLOO_SOCKET_IFNAME", None)
            elif had:
                os.environ["GLOO_SOCKET_IFNAME"] = prev
    return _HOST_GROUPS[be]


class DPContext(object):
    def __init__(self, world=1, rank=0, group=None):
        self.world, self.rank, self.group = int(world), int(rank), group
        self._host = None

    @classmethod
    def from_env(cls):
        """Collective when a process group exists (the first call creates the host-side gloo group): every rank must call it
        at the same point - the trainers do, in their constructors / make_optimizers()."""
        if dist.is_available() and dist.is_initialized():
            c = cls(dist.get_world_size(), dist.get_rank())
            if c.world > 1:
                c._host = _host_group()
            return c
        return cls(1, 0)

    @property
    def active(self):
        # AAS_DP_FORCE=1: run the data-parallel code path (collectives, global normalisers, bucket reducer) on a one-rank
        # group too - how the RCCL calls are exercised on a single-GPU test box
        return self.world > 1 or (os.environ.get("AAS_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized())

    def _dev(self, like=None):
        if like is not None:
            return like.device
        return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(self.group) == "nccl" else torch.device("cpu")

    def _host_ready(self):
        return self._host is not None or dist.get_backend(self.group) == "gloo"

    def barrier(self):
        """Host-side barrier.  Over the gloo side group when the data path is RCCL: it has no device work, so it neither
        trips RCCL's watchdog (default 10 minutes) while rank 0 validates nor orders itself against queued collectives."""
        if not self.active:
            return
        if self._host_ready():
            dist.barrier(group=self._host if self._host is not None else self.group)
        else:
            dist.barrier(group=self.group)

    def host_max(self, value):
        """MAX over ranks of one host integer (the padded length of a sharded batch) without touching the device."""
        if not self.active:
            return int(value)
        if self._host_ready():
            t = torch.tensor([int(value)], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self._host if self._host is not None else self.group)
            return int(t[0])
        t = torch.tensor([int(value)], dtype=torch.int64, device=self._dev())
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return int(t.item())

    def global_counts(self, values):
        """SUM over ranks of a short list of integers (N, nElement ...) -> list[int]."""
        if not self.active:
            return [int(v) for v in values]
        t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=self._dev())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return [int(round(v)) for v in t.tolist()]

    def _staged(self, tensor):
        """gloo cannot reduce device tensors on every build: stage through the host (tests only; the RCCL path
        reduces in place on the device)."""
        return tensor.is_cuda and dist.get_backend(self.group) == "gloo"

    def allreduce_sum_(self, tensor, async_op=False):
        """In-place SUM all-reduce of a (flat gradient) buffer; returns a work handle if async."""
        if not self.active:
            return None
        if self._staged(tensor):
            host = tensor.detach().cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            tensor.copy_(host)
            return _Done()
        if tensor.is_cuda:
            from . import ops
            # (ops.Profiler class "coll": HIP events on the issuing stream around the call - WHEN a bucket's collective becomes
            #  eligible relative to the step's other launches; tools/event_timeline.py --classes rnn,gemm,coll)
            with ops._timed("coll", "allreduce[%.1f MB]" % (tensor.numel() * tensor.element_size() / 2.0 ** 20), 0.0):
                return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        return dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def broadcast_(self, tensor, src=0):
        """In-place broadcast from rank `src` (module buffers that only one rank advanced: A's BatchNorm running statistics behind
        a rank-0-only validation pass)."""
        if not self.active or self.world == 1:
            return tensor
        if self._staged(tensor):
            host = tensor.detach().cpu()
            dist.broadcast(host, src=src, group=self.group)
            tensor.copy_(host)
        else:
            dist.broadcast(tensor, src=src, group=self.group)
        return tensor

    def broadcast_buffers(self, module, src=0):
        """Every registered buffer of `module` from rank `src`: one flat float broadcast + one for the integer counters."""
        if not self.active or self.world == 1:
            return
        fl = [b for b in module.buffers() if b.dtype.is_floating_point]
        it = [b for b in module.buffers() if not b.dtype.is_floating_point]
        for group in (fl, it):
            if not group:
                continue
            flat = torch.cat([b.detach().reshape(-1).to(group[0].dtype) for b in group])
            self.broadcast_(flat, src)
            off = 0
            for b in group:
                n = b.numel()
                b.copy_(flat[off:off + n].view_as(b))
                off += n

    def reduce_scalars(self, tensor):
        if self.active:
            if self._staged(tensor):
                host = tensor.detach().cpu()
                dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                tensor.copy_(host)
            else:
                dist.all_reduce(tensor, op=dist.ReduceOp.SUM, group=self.group)
        return tensor

    # ---- batch sharding ----------------------------------------------------------------------
    def shard_rows(self, n):
        return list(range(self.rank, n, self.world))

    def shard_collated(self, data_list):
        """Shard an already collated GLOBAL `_collate_fn` tuple (inputs, targets, pct, target_sizes, mask) by stride - for callers
        that hold a whole batch (tests, synthetic benchmarks).  The loaders shard BEFORE loading instead
        (loader_functions.FeatSampler(rank, world) + data_loader.DataLoader(dp=...)): a rank then never opens another rank's files.
        The number of valid frames comes from the host-side `input_percentages` (no device read-back)."""
        if not self.active:
            return data_list
        inputs, targets, pct, tsz, mask = data_list
        if inputs.size(0) < self.world:
            raise ValueError("data parallel: a global batch of %d utterances cannot be sharded over %d ranks (every rank needs at "
                             "least one row; FeatSampler(world=...) drops such bins identically on every rank)" % (inputs.size(0), self.world))
        rows = self.shard_rows(inputs.size(0))
        idx = torch.tensor(rows, dtype=torch.long)
        offs = [0]
        for s in tsz.tolist():
            offs.append(offs[-1] + int(s))
        tg = torch.cat([targets[offs[r]:offs[r + 1]] for r in rows]) if targets is not None and len(rows) else targets
        m = mask.index_select(0, idx.to(mask.device))
        if pct is not None and not pct.is_cuda:
            # T_i = pct_i * T_max exactly as _collate_fn formed it (lengths / T_max in fp64, rounded to fp32)
            m.n_valid = int(torch.round(pct.index_select(0, idx).double() * inputs.size(2)).sum().item())
        else:
            m.n_valid = int(m.numel()) - int(m.sum().item())     # (no host lengths; a device mask costs one read-back)
        return (inputs.index_select(0, idx.to(inputs.device)), tg, pct.index_select(0, idx), tsz.index_select(0, idx), m)


class DeviceCounts(object):
    """SUM over ranks of a few per-rank counts (N, nElement ...) WITHOUT a host synchronisation: the values go up through a
    pinned buffer, are all-reduced on an auxiliary stream and stay on the device as fp64 scalars; a stream that reads one
    waits for that stream's event.  (global_counts() returns python ints instead and therefore blocks the host until the
    device has caught up - fine for a logging path, not at the top of a training step.)"""

    def __init__(self, dp, values, device, aux_stream):
        host = torch.tensor([float(v) for v in values], dtype=torch.float64)
        if not os.environ.get("AAS_NO_PIN"):
            host = host.pin_memory()     # a fresh pinned buffer per call (a few bytes): the copy below never blocks the host and no
                                         # later call can overwrite it before it has run
        main = torch.cuda.current_stream()
        aux_stream.wait_stream(main)
        with torch.cuda.stream(aux_stream):
            self.t = host.to(device, non_blocking=True)
            dp.reduce_scalars(self.t)
            self.ev = torch.cuda.Event()
            self.ev.record(aux_stream)
        self.t.record_stream(main)

    def get(self, i):
        torch.cuda.current_stream().wait_event(self.ev)
        return self.t[i]


class DeviceScales(object):
    """Loss normalisers of a data-parallel step that never leave the device: `cnt` (a device fp64 vector of this rank's counts - N,
    nElement ... - uploaded by the caller) is SUM-all-reduced on the auxiliary stream, then ONE library launch forms
    scales[i] = weights[i] / cnt[index[i]] as device floats (include/aas_hip.h: aas_scales_from_counts).  `sc[i]` / `sc.count(i)`
    make the CURRENT stream wait for that stream's event (device side, not host side) and return a 0-dim device view: the
    collective's latency hides behind whatever the caller queued in between."""

    def __init__(self, dp, cnt, weights, index, aux):
        from . import ops
        main = torch.cuda.current_stream()
        aux.wait_stream(main)                      # the upload of `cnt`
        with torch.cuda.stream(aux):
            dp.reduce_scalars(cnt)
            self.all = torch.empty(max(3, len(weights)), device=cnt.device, dtype=torch.float32)
            ops.scales_from_counts(cnt, list(weights), list(index), self.all)
            self.vals = tuple(self.all[i] for i in range(len(weights)))
            self.cnt = cnt
            self.ev = torch.cuda.Event()
            self.ev.record(aux)
        for t_ in (self.all, cnt):
            t_.record_stream(main)

    def __getitem__(self, i):
        torch.cuda.current_stream().wait_event(self.ev)
        return self.vals[i]

    def count(self, i):
        torch.cuda.current_stream().wait_event(self.ev)
        return self.cnt[i]


class _Done(object):
    def wait(self):
        return True


class FlatBuffers(object):
    """Parameters and gradients of a module re-homed into two flat fp32 buffers (views keep the
    nn.Parameter API and state_dict keys intact): ONE Adam launch, ONE grad-norm launch and ONE
    all-reduce per network instead of one per tensor."""

    def __init__(self, module):
        self.params = [p for p in module.parameters()]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat_p = torch.empty(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        self.slices = []
        for p in self.params:
            n = p.numel()
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view_as(p)
            self.slices.append((off, n))
            off += n
            # marks the parameter for ops._BiRNNLayer: its weight gradients may be accumulated straight into .grad on the
            # side stream, because whoever owns these buffers joins that stream (ops.sync_wgrad) before reading them
            p._aas_flat_grad = True
            p._aas_flat_ref = self      # ops caches weight operand planes per (address, version, self.version)
        self.version = 0                # bumped by optim.FlatAdam: its HIP kernel does not touch torch's version counters
        self.bind_grads()

    def bind_grads(self):
        for p, (off, n) in zip(self.params, self.slices):
            g = self.flat_g[off:off + n].view_as(p)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def zero_grad(self):
        self.flat_g.zero_()
        self.bind_grads()


class BucketReducer(object):
    """Bucketed, overlapped SUM all-reduce of the flat gradient buffers (>= 16 MB buckets launched as each
    network's weight gradients finish, D's first, then E's).

    A recurrent layer's four weight-gradient products are queued on the side stream by ops._birnn_bwd, which then calls
    `on_wgrad(grads)` from that stream's context: the layer's contiguous slice of the flat buffer (16 MB for a 500-unit
    BiLSTM layer, 48 MB for a 1000-unit BiGRU layer) is all-reduced right there, ordered after the products on the side
    stream and overlapping the rest of the backward pass.  `flush(flat)` reduces whatever the hooks did not cover (the
    small pointwise / conv / BN / fc parameters) and `wait()` makes the current stream wait for every collective.
    Armed only in steps where every layer is back-propagated exactly once (the fused schedule without grad-norm logging)."""
    MIN_ELEMS = 1 << 20

    def __init__(self, dp, flats):
        self.dp, self.flats = dp, list(flats)
        self.handles, self.done = [], {}
        self.deferred = set()

    def defer_layer(self, grads):
        """The layer with these gradients is the LAST one of a step's backward pass: its bucket would be followed at once by
        flush() of its neighbours in the buffer, so it is left to flush() - one collective for the contiguous range instead
        of two or three small ones at the exposed end of the step."""
        f, lo, hi, n = self._locate(grads)
        if f is not None:
            self.deferred.add((id(f), lo))

    def begin(self):
        self.handles = []
        self.done = {id(f): [] for f in self.flats}
        self.expected, self.seen = {}, {}

    def expect(self, flat, times):
        """Layers of `flat` are back-propagated `times` times this step (two D passes): reduce a bucket after the last one."""
        self.expected[id(flat)] = int(times)

    def _locate(self, grads):
        for f in self.flats:
            base, total = f.flat_g.data_ptr(), f.flat_g.numel()
            offs = [(g.data_ptr() - base) // 4 for g in grads]
            if all(0 <= o < total for o in offs):
                lo = min(offs)
                hi = max(o + g.numel() for o, g in zip(offs, grads))
                return f, lo, hi, sum(g.numel() for g in grads)
        return None, 0, 0, 0

    def on_wgrad(self, grads):
        f, lo, hi, n = self._locate(grads)
        if f is None or hi - lo != n or n < self.MIN_ELEMS:
            return
        k = (id(f), lo)
        if k in self.deferred:
            return
        self.seen[k] = self.seen.get(k, 0) + 1
        if self.seen[k] < self.expected.get(id(f), 1):
            return
        self.handles.append(self.dp.allreduce_sum_(f.flat_g[lo:hi], async_op=True))
        self.done[id(f)].append((lo, hi))

    def flush(self, flat):
        """Reduce every range of `flat` no bucket covered.  Issued from the weight-gradient stream (behind the products
        queued there so far, which may include these ranges when a layer was too small for a bucket) after an event of the
        calling stream (where autograd accumulated the small parameters' gradients): the caller's stream is not held up."""
        def issue():
            pos = 0
            for lo, hi in sorted(self.done[id(flat)]) + [(flat.flat_g.numel(), flat.flat_g.numel())]:
                if lo > pos:
                    self.handles.append(self.dp.allreduce_sum_(flat.flat_g[pos:lo], async_op=True))
                pos = max(pos, hi)
            self.done[id(flat)] = [(0, flat.flat_g.numel())]
        if flat.flat_g.is_cuda:
            from . import ops
            side = ops.wgrad_stream(flat.flat_g.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                side.wait_event(ev)
                issue()
        else:
            issue()

    def wait(self):
        for h in self.handles:
            if h is not None:
                h.wait()
        self.handles = []
