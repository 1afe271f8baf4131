"""Stage-to-stage exchange of activation / gradient tuples.

On MI355X each neighbour pair talks over its direct xGMI link through RCCL point-to-point (torch.distributed
backend 'nccl').  All P2P traffic is issued on one side HIP stream, ordered against the compute stream with
events: a send waits only for the kernel that produced its payload, a receive is ordered before the first kernel
that consumes it, so transfers overlap with the compute of other micro-batches.  The per-step instruction order of
the 1F1B schedule makes adjacent stages post complementary send/recv sequences, which keeps the single in-order
communication stream deadlock-free.  On CPU tensors (gloo, used by the tests) the same code runs without streams.

Wire format (dynamic shapes, like DeepSpeed's `_send_tensor_meta`): the first tuple sent across a boundary after
`reset_activation_shape()` is preceded by one int64 header [n, (dtype_id, ndim, *shape) * n]; later tuples reuse it.
"""
import torch
import torch.distributed as dist

_DTYPES = [torch.float32, torch.bfloat16, torch.float16, torch.float64, torch.int64, torch.int32, torch.int16, torch.int8,
           torch.uint8, torch.bool, torch.complex64, torch.complex128]
_DTYPE_ID = {dt: i for i, dt in enumerate(_DTYPES)}
_MAX_META = 512


def encode_meta(tensors, is_tuple=True):
    meta = [len(tensors), int(is_tuple)]
    for t in tensors:
        meta += [_DTYPE_ID[t.dtype], t.dim(), *t.shape]
    if len(meta) + 1 > _MAX_META:
        raise RuntimeError('activation tuple too large for the P2P header')
    return [len(meta)] + meta + [0] * (_MAX_META - 1 - len(meta))


def decode_meta(words):
    n, is_tuple = words[1], bool(words[2])
    pos = 3
    out = []
    for _ in range(n):
        dtype, ndim = _DTYPES[words[pos]], words[pos + 1]
        shape = tuple(words[pos + 2: pos + 2 + ndim])
        pos += 2 + ndim
        out.append((dtype, shape))
    return out, is_tuple


def _wire(t):
    """View of `t` in a dtype every backend can move (RCCL has no complex / bool types)."""
    if t.is_complex():
        return torch.view_as_real(t)
    if t.dtype == torch.bool:
        return t.view(torch.uint8)
    return t


class StageLink:
    """P2P endpoint of one pipeline stage."""

    def __init__(self, grid, device, comm_stream=None):
        self.grid = grid
        self.device = torch.device(device)
        self.on_gpu = self.device.type == 'cuda'
        # `comm_stream`: the engine hands over a stream probed to sit on a hardware queue of its own (engine.concurrent_streams) BEFORE the link runs anything on it
        self.comm_stream = comm_stream if comm_stream is not None else (torch.cuda.Stream(self.device) if self.on_gpu else None)
        self._pending = []          # (work, tensors) of sends not yet waited for
        self.reset()

    def reset(self):
        """Forget cached tuple layouts (engine.reset_activation_shape)."""
        self._sent_meta = {}        # (peer, tag) -> layout already announced
        self._recv_meta = {}        # (peer, tag) -> layout received

    # ------------------------------------------------------------------------------------------ raw transfers
    def _isend(self, tensors, peer):
        if self.on_gpu:
            ev = torch.cuda.current_stream(self.device).record_event()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                works = [dist.isend(_wire(t), dst=peer) for t in tensors]
            for t in tensors:
                t.record_stream(self.comm_stream)
        else:
            works = [dist.isend(_wire(t), dst=peer) for t in tensors]
        self._pending.append((works, tensors))

    @torch.no_grad()            # the buffers may be a stage graph's static inputs (leaves that require grad)
    def _fresh(self, make):
        """Receive buffers nobody owns yet, allocated ON THE COMMUNICATION STREAM: the caching allocator recycles a block only for the stream it was
        freed on, so a block handed out here was last used by this link's own transfers (or by a consumer whose use `record_stream` registered) -- the receive needs no
        wait on the caller's stream, i.e. it is NOT serialised behind the compute already enqueued there (round 4 ordered every fresh receive behind it)."""
        if self.on_gpu and self.comm_stream is not None:
            with torch.cuda.stream(self.comm_stream):
                return make(), True
        return make(), False

    def _recv(self, buffers, peer, fresh=True, owned=False):
        if self.on_gpu and fresh and not owned:
            # buffers allocated on the CALLER'S stream may be a block the caching allocator just recycled there: kernels already enqueued can still read its previous
            # contents, so the transfer is ordered behind them (receives into a slot's static buffers are ordered by their own `after` event instead)
            self.comm_stream.wait_event(torch.cuda.current_stream(self.device).record_event())
        if self.on_gpu:
            with torch.cuda.stream(self.comm_stream):
                works = [dist.irecv(_wire(b), src=peer) for b in buffers]
                for w in works:
                    w.wait()                          # comm stream ordered after the transfers
                done = self.comm_stream.record_event()
            cur = torch.cuda.current_stream(self.device)
            for b in buffers:
                b.record_stream(cur if owned else self.comm_stream)      # the stream that did NOT allocate the block uses it too
            cur.wait_event(done)   # consumers run after the payload landed
        else:
            for w in [dist.irecv(_wire(b), src=peer) for b in buffers]:
                w.wait()

    def flush(self):
        """Complete outstanding sends (end of a batch)."""
        for works, _ in self._pending:
            for w in works:
                w.wait()
        self._pending.clear()

    # ------------------------------------------------------------------------------------------ tuple protocol
    def send_tuple(self, payload, peer_stage, tag):
        """payload: a tensor or a tuple/list of tensors (the value a layer returned)."""
        peer = self.grid.stage_to_global(peer_stage)
        is_tuple = not torch.is_tensor(payload)
        tensors = list(payload) if is_tuple else [payload]
        tensors = [t if t.is_contiguous() else t.contiguous() for t in tensors]
        layout = ([(t.dtype, tuple(t.shape)) for t in tensors], is_tuple)
        key = (peer, tag)
        if self._sent_meta.get(key) != layout:
            if key in self._sent_meta:
                raise RuntimeError('activation shapes changed without reset_activation_shape()')
            header = torch.tensor(encode_meta(tensors, is_tuple), dtype=torch.int64, device=self.device)
            self._isend([header], peer)
            self._sent_meta[key] = layout
        if tensors:
            self._isend(tensors, peer)

    def recv_tuple(self, peer_stage, tag, into=None, after=None):
        """into: buffers to receive into when their layout equals the announced one (a stage graph's static inputs: no staging copy);
        after: event the receive must wait for before it may overwrite them."""
        peer = self.grid.stage_to_global(peer_stage)
        key = (peer, tag)
        layout = self._recv_meta.get(key)
        if layout is None:
            header = torch.empty(_MAX_META, dtype=torch.int64, device=self.device)
            self._recv([header], peer)
            layout = decode_meta(header.tolist())     # host read: once per boundary per shape epoch
            self._recv_meta[key] = layout
        specs, is_tuple = layout
        direct = into is not None and len(into) == len(specs) and all(
            b.dtype == dtype and tuple(b.shape) == tuple(shape) and b.is_contiguous() for b, (dtype, shape) in zip(into, specs))
        owned = False
        if direct:
            buffers = list(into)
        else:
            buffers, owned = self._fresh(lambda: [torch.empty(shape, dtype=dtype, device=self.device) for dtype, shape in specs])
        if buffers:
            self._recv_after(buffers, peer, after if direct else None, owned=owned)
        return tuple(buffers) if is_tuple else buffers[0]

    def _recv_after(self, buffers, peer, after, owned=False):
        if after is None:
            return self._recv(buffers, peer, owned=owned)
        if isinstance(self, RcclLink):
            return self._recv(buffers, peer, after)
        if self.comm_stream is not None:
            self.comm_stream.wait_event(after)
            torch.cuda.current_stream(self.device).wait_event(after)     # host-staged endpoints copy on the current stream
        return self._recv(buffers, peer, fresh=False)

    def recv_like(self, templates, peer_stage, into=None, after=None):
        """Receive tensors whose layouts are known locally (gradients of tensors this stage sent)."""
        peer = self.grid.stage_to_global(peer_stage)
        direct = into is not None and len(into) == len(templates) and all(
            b.dtype == t.dtype and b.shape == t.shape and b.is_contiguous() for b, t in zip(into, templates))
        owned = False
        if direct:
            buffers = list(into)
        else:
            buffers, owned = self._fresh(lambda: [torch.empty_like(t, memory_format=torch.contiguous_format) for t in templates])
        if buffers:
            self._recv_after(buffers, peer, after if direct else None, owned=owned)
        return buffers

    def send_plain(self, tensors, peer_stage):
        peer = self.grid.stage_to_global(peer_stage)
        tensors = [t if t.is_contiguous() else t.contiguous() for t in tensors]
        if tensors:
            self._isend(tensors, peer)


class RcclLink(StageLink):
    """P2P endpoint on the C-ABI RCCL wrappers (csrc/comm.hip: dpipe_send / dpipe_recv) instead of torch.distributed's
    isend / irecv: one 2-rank communicator per neighbour stage, every tuple leaves as ONE grouped RCCL operation on the communication stream
    (the `batch_isend_irecv` shape), and a receive lands directly in the buffers the caller names (`into=`: a stage graph's static inputs)
    with no staging copy.  Communicator ids travel over the pipeline's torch.distributed group once, at construction.
    Engine config `p2p_backend`: 'rccl' = this link, 'torch' = StageLink, 'auto' (default on GPU) = this link when its construction + self-test
    succeed on every rank of the world, else StageLink."""

    def __init__(self, grid, device, self_test=True, comm_stream=None, connect=True):
        super().__init__(grid, device, comm_stream=comm_stream)
        from .. import hip
        self._hip = hip
        self._comms = {}            # peer global rank -> (comm handle, peer's rank inside the pair communicator)
        if connect:
            err = self._connect()
            if err is not None:
                raise err
            if self_test:
                err = self._self_test()
                if err is not None:
                    raise err

    last_negotiation = None

    @classmethod
    def negotiate(cls, grid, device, comm_stream=None, agree=None, log=print):
        """All-ranks-agree construction for `p2p_backend: 'auto'`: returns a connected, self-tested link on EVERY rank of the world or None on every rank.

        The decision is taken in phases, each closed by a world all-reduce(MIN) of a local ok flag, and no rank ever leaves a phase early (ADVICE round 3:
        a rank that raised out of `_connect` went straight to the agreement while its neighbours still sat in the pipe-group broadcast / ncclCommInitRank /
        a self-test receive -- a deadlock instead of the documented fallback):
          0. local probe, no collectives: the library loads, RCCL resolves and a ncclUniqueId can be drawn;
          1. `_connect`: every stage takes part in every broadcast of its pipe group (a lower stage that cannot draw an id broadcasts None and the pair
             skips its rendezvous together); communicator errors are recorded, not raised;
          2. `_self_test`: a corrupted pattern is recorded and the chain of sends still completes, so no neighbour is left in a receive."""
        import ctypes
        device = torch.device(device)
        if agree is None:
            def agree(ok):
                flag = torch.tensor([1 if ok else 0], device=device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                return int(flag.item()) == 1
        rank = grid.global_rank

        cls.last_negotiation = {'link': 'RcclLink', 'failed_phase': None, 'local_error': None}      # per process: what THIS rank saw (bench.py gathers it per rank)

        def phase(name, err):
            if err is not None:
                cls.last_negotiation['local_error'] = f'{name}: {err!r}'[:200]
                log(f'[dpipe] rank {rank}: RCCL link unavailable at {name} ({err}); stage exchange falls back to torch.distributed isend / irecv', flush=True)
            ok = agree(err is None)
            if not ok:
                cls.last_negotiation.update(link='StageLink (torch isend / irecv)', failed_phase=name)
            return ok

        err, link = None, None
        try:
            from .. import hip
            buf = ctypes.create_string_buffer(128)
            hip.check(hip.lib().dpipe_comm_unique_id(buf), 'comm_unique_id')
        except Exception as e:                            # noqa: BLE001  (library / RCCL symbols missing, id not drawable)
            err = e
        if not phase('probe', err):
            return None
        try:
            link = cls(grid, device, comm_stream=comm_stream, connect=False)
            err = link._connect()
        except Exception as e:                            # noqa: BLE001
            err = e
        if not phase('connect', err):
            if link is not None:
                link.close()
            return None
        try:
            err = link._self_test()
        except Exception as e:                            # noqa: BLE001
            err = e
        if not phase('self-test', err):
            link.close()
            return None
        return link

    def close(self):
        """Destroy this endpoint's communicators (a link the ranks agreed not to use)."""
        for comm, _ in self._comms.values():
            try:
                self._hip.lib().dpipe_comm_destroy(comm)
            except Exception:                             # noqa: BLE001
                pass
        self._comms = {}

    def _connect(self):
        """One 2-rank RCCL communicator per neighbour pair of THIS pipeline, created eagerly and in stage order by every rank of the pipeline: the
        lower stage draws the ncclUniqueId and broadcasts it over the pipeline's existing process group (a collective every stage of the pipeline
        takes part in, so no extra groups exist and ranks never reach a rendezvous at different points of the schedule); the two neighbours then
        initialise their communicator.  With data parallelism the pipelines are disjoint rank sets (rank = stage * dp + replica), each does this
        over its own pipe group.  Never raises and never skips a broadcast: returns the first error (or None) once every pair has been visited."""
        import ctypes
        grid = self.grid
        stages, me = grid.pipe_parallel_size, grid.global_rank
        group = grid.get_pipe_parallel_group()
        first_err = None
        for s in range(stages - 1):
            lo, hi = grid.stage_to_global(s), grid.stage_to_global(s + 1)
            box = [None]
            if me == lo:
                try:
                    buf = ctypes.create_string_buffer(128)
                    self._hip.check(self._hip.lib().dpipe_comm_unique_id(buf), 'comm_unique_id')
                    box[0] = buf.raw
                except Exception as e:                    # noqa: BLE001 -- broadcast the sentinel instead: both ends skip the rendezvous together
                    first_err = first_err or e
            dist.broadcast_object_list(box, src=lo, group=group)
            if me in (lo, hi):
                if box[0] is None:
                    first_err = first_err or RuntimeError(f'stage {s} could not draw a ncclUniqueId for the pair ({s}, {s + 1})')
                    continue
                try:
                    comm = ctypes.c_void_p()
                    with torch.cuda.device(self.device):
                        self._hip.check(self._hip.lib().dpipe_comm_init(ctypes.byref(comm), 2, 0 if me == lo else 1, box[0]), 'comm_init')
                    self._comms[hi if me == lo else lo] = (comm, 1 if me == lo else 0)
                except Exception as e:                    # noqa: BLE001
                    first_err = first_err or e
        return first_err

    def _self_test(self):
        """A 4 KiB pattern travels down the pipeline and back over every communicator before the first real tuple does.  A corrupted pattern is RECORDED and
        the chain goes on (the next stage still gets its send), so a failure never strands a neighbour in a receive; returns the first error or None."""
        grid = self.grid
        s, S = grid.get_stage_id(), grid.pipe_parallel_size
        want = torch.arange(1024, dtype=torch.int32, device=self.device)
        buf = torch.empty_like(want)
        err = None
        if s > 0:
            self._recv([buf], grid.stage_to_global(s - 1))
            torch.cuda.current_stream(self.device).synchronize()
            if not torch.equal(buf, want + (s - 1)):
                err = err or RuntimeError(f'RcclLink self-test: stage {s} received a corrupted pattern from stage {s - 1}')
        if s < S - 1:
            self._isend([want + s], grid.stage_to_global(s + 1))
            self._recv([buf], grid.stage_to_global(s + 1))
            torch.cuda.current_stream(self.device).synchronize()
            if not torch.equal(buf, want - (s + 1)):
                err = err or RuntimeError(f'RcclLink self-test: stage {s} received a corrupted pattern from stage {s + 1}')
        if s > 0:
            self._isend([want - s], grid.stage_to_global(s - 1))
        self.flush()
        torch.cuda.current_stream(self.device).synchronize()
        return err

    def _comm(self, peer):
        hit = self._comms.get(peer)
        if hit is None:
            raise RuntimeError(f'rank {self.grid.global_rank} has no RCCL communicator with rank {peer}: stages exchange tuples with their pipeline neighbours only')
        return hit

    def _grouped(self, tensors, peer, op):
        comm, peer_rank = self._comm(peer)
        lib, st = self._hip.lib(), self.comm_stream.cuda_stream
        self._hip.check(lib.dpipe_group_start(), 'group_start')
        try:
            for t in tensors:
                w = _wire(t)
                self._hip.check(op(lib)(comm, w.data_ptr(), w.numel() * w.element_size(), peer_rank, st), 'send / recv')
        finally:
            rc = lib.dpipe_group_end()              # always close the group: an open RCCL group would swallow every later call of this thread
        self._hip.check(rc, 'group_end')

    def _isend(self, tensors, peer):
        ev = torch.cuda.current_stream(self.device).record_event()
        self.comm_stream.wait_event(ev)
        self._grouped(tensors, peer, lambda lib: lib.dpipe_send)
        for t in tensors:
            t.record_stream(self.comm_stream)
        self._pending.append(([], tensors))              # kept alive until flush(); completion is stream-ordered

    def _recv(self, buffers, peer, after=None, owned=False):
        if after is not None:
            self.comm_stream.wait_event(after)           # the buffers' previous consumer (a stage graph replay) has finished
        elif not owned:                                  # buffers allocated on the caller's stream: behind whatever it still runs on the recycled block (see StageLink._recv);
            self.comm_stream.wait_event(torch.cuda.current_stream(self.device).record_event())      # `owned` = allocated on the communication stream (StageLink._fresh): no wait
        self._grouped(buffers, peer, lambda lib: lib.dpipe_recv)
        done = self.comm_stream.record_event()
        cur = torch.cuda.current_stream(self.device)
        for b in buffers:
            b.record_stream(cur if owned else self.comm_stream)
        cur.wait_event(done)

    def flush(self):
        if self._pending:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        self._pending.clear()


class HostStagedLink(StageLink):
    """Test / fallback endpoint: payloads cross through host memory over a CPU-capable backend (gloo).  This is how the
    reference runs consumer cards (`NCCL_P2P_DISABLE=1`, README.md:118-120) and what lets the pipeline-parallel engine
    (including its per-stage hipGraphs) run as two processes sharing ONE GPU in the GPU tests; the MI355X product path
    is `StageLink` (RCCL over xGMI)."""

    def _isend(self, tensors, peer):
        torch.cuda.current_stream(self.device).synchronize()
        host = [_wire(t).detach().cpu() for t in tensors]
        works = [dist.isend(h, dst=peer) for h in host]
        self._pending.append((works, host))

    def _fresh(self, make):
        return make(), False                 # the staging copy runs on the caller's stream: allocate there

    def _recv(self, buffers, peer, fresh=True, owned=False):
        for b in buffers:
            w = _wire(b)
            h = torch.empty(w.shape, dtype=w.dtype)
            dist.recv(h, src=peer)
            with torch.no_grad():
                w.copy_(h)
