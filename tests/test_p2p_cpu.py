"""Stage-to-stage exchange protocol (engine/p2p.py) on CPU / gloo: header encode / decode round trip for every wire
dtype, tuples vs single tensors, dynamic shapes (one header per boundary per shape epoch, error on a silent shape change), bool / complex
payloads through their wire views, gradient receive by template."""
import os
import socket
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusion_pipe_amd.engine import p2p


def test_header_round_trip_all_dtypes():
    tensors = [torch.zeros(s, dtype=dt) for dt, s in zip(p2p._DTYPES, [(2, 3), (1,), (4, 1, 2), (), (5,), (2, 2), (3,), (1, 1), (7,), (2,), (3, 2), (1, 2, 3)])]
    for is_tuple in (True, False):
        words = p2p.encode_meta(tensors if is_tuple else tensors[:1], is_tuple)
        assert len(words) == p2p._MAX_META and words[0] <= p2p._MAX_META - 1
        specs, got_tuple = p2p.decode_meta(words)
        want = tensors if is_tuple else tensors[:1]
        assert got_tuple == is_tuple and specs == [(t.dtype, tuple(t.shape)) for t in want]
    with pytest.raises(RuntimeError, match='too large'):
        p2p.encode_meta([torch.zeros((1,) * 8)] * 60)


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _worker(rank, port, outdir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE='2')
    dist.init_process_group('gloo', rank=rank, world_size=2)
    try:
        grid = type('G', (), {'stage_to_global': staticmethod(lambda s: s)})
        link = p2p.StageLink(grid, 'cpu')
        g = torch.Generator().manual_seed(3)
        payloads = [
            (torch.randn(2, 5, generator=g), torch.arange(6).view(2, 3), torch.randn(2, 5, generator=g).to(torch.bfloat16)),     # mixed tuple
            torch.randn(4, 4, generator=g),                                                                                      # single tensor
            (torch.tensor([True, False, True]), torch.complex(torch.randn(3, generator=g), torch.randn(3, generator=g))),         # bool + complex wire views
            (torch.randn(6, 2, generator=g).t(),),                                                                               # non-contiguous -> contiguous on the wire
        ]
        log = {}
        if rank == 0:
            for i, p in enumerate(payloads):
                link.send_tuple(p, 1, tag=f'b{i}')
            link.send_tuple(payloads[1] * 2, 1, tag='b1')                       # same boundary, same layout: no second header
            headers_sent = len(link._sent_meta)
            try:
                link.send_tuple(torch.zeros(3, 3), 1, tag='b1')                 # layout changed without reset_activation_shape()
                log['shape_change'] = 'no error'
            except RuntimeError as e:
                log['shape_change'] = str(e)
            link.reset()
            link.send_tuple(torch.zeros(3, 3), 1, tag='b1')                     # after reset: a new header goes out
            grads = link.recv_like([torch.empty(2, 5), torch.empty(4, 4, dtype=torch.bfloat16)], 1)
            link.flush()
            log.update(headers_sent=headers_sent, grads=[x.clone() for x in grads])
        else:
            got = [link.recv_tuple(0, tag=f'b{i}') for i in range(len(payloads))]
            again = link.recv_tuple(0, tag='b1')
            link.reset()
            after_reset = link.recv_tuple(0, tag='b1')
            link.send_plain([torch.full((2, 5), 7.0), torch.full((4, 4), 3.0, dtype=torch.bfloat16)], 0)
            link.flush()
            log.update(got=got, again=again, after_reset=after_reset)
        torch.save(log, os.path.join(outdir, f'r{rank}.pt'))
    finally:
        dist.destroy_process_group()


def test_tuple_protocol_over_gloo():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(_free_port(), d), nprocs=2, join=True)
        send, recv = (torch.load(os.path.join(d, f'r{r}.pt'), weights_only=False) for r in range(2))
    g = torch.Generator().manual_seed(3)
    a, c = torch.randn(2, 5, generator=g), torch.randn(2, 5, generator=g).to(torch.bfloat16)
    got = recv['got']
    assert isinstance(got[0], tuple) and torch.equal(got[0][0], a) and torch.equal(got[0][1], torch.arange(6).view(2, 3)) and torch.equal(got[0][2], c)
    single = torch.randn(4, 4, generator=g)
    assert torch.is_tensor(got[1]) and torch.equal(got[1], single) and torch.equal(recv['again'], single * 2)
    assert got[2][0].dtype == torch.bool and got[2][0].tolist() == [True, False, True] and got[2][1].dtype == torch.complex64
    assert got[3][0].shape == (2, 6) and got[3][0].is_contiguous()
    assert send['headers_sent'] == 4 and 'shapes changed' in send['shape_change']
    assert recv['after_reset'].shape == (3, 3)
    assert torch.equal(send['grads'][0], torch.full((2, 5), 7.0)) and send['grads'][1].dtype == torch.bfloat16


def _negotiate_worker(rank, port, outdir, world, fail_rank, fail_phase):
    """RcclLink.negotiate over gloo with the native calls stubbed: one rank fails at one phase, every rank must still reach every agreement."""
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import datetime
        group = dist.new_group(list(range(world)), timeout=datetime.timedelta(seconds=60))
        grid = type('G', (), {'stage_to_global': staticmethod(lambda s: s), 'pipe_parallel_size': world, 'global_rank': rank,
                              'get_pipe_parallel_group': staticmethod(lambda: group), 'get_stage_id': staticmethod(lambda: rank)})
        calls = []

        class FakeLib:
            def dpipe_comm_unique_id(self, buf):
                calls.append('id')
                return -1 if (rank == fail_rank and fail_phase in ('probe', 'connect-id') and (fail_phase == 'probe' or len([c for c in calls if c == 'id']) > 1)) else 0

            def dpipe_comm_init(self, comm, world_, r, id_):
                calls.append('init')
                return -1 if (rank == fail_rank and fail_phase == 'connect-init') else 0

            def dpipe_comm_destroy(self, comm):
                calls.append('destroy')
                return 0

        class FakeHip:
            @staticmethod
            def lib():
                return FakeLib()

            @staticmethod
            def check(rc, what):
                if rc != 0:
                    raise RuntimeError(f'{what} failed')

        import sys
        import types
        import diffusion_pipe_amd
        fake = types.ModuleType('diffusion_pipe_amd.hip')
        fake.lib, fake.check = FakeHip.lib, FakeHip.check
        sys.modules['diffusion_pipe_amd.hip'] = fake
        diffusion_pipe_amd.hip = fake

        class Link(p2p.RcclLink):
            def _self_test(self):            # the data path needs a GPU; the phase protocol does not
                calls.append('selftest')
                return RuntimeError('corrupted pattern') if (rank == fail_rank and fail_phase == 'self-test') else None

        class Dev:
            def __enter__(self): return self
            def __exit__(self, *a): return False
        orig = torch.cuda.device
        torch.cuda.device = lambda d: Dev()
        try:
            link = Link.negotiate(grid, 'cpu', log=lambda *a, **k: None)
        finally:
            torch.cuda.device = orig
        torch.save({'link': link is not None, 'calls': calls}, os.path.join(outdir, f'n{rank}.pt'))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('fail_phase', ['none', 'probe', 'connect-id', 'connect-init', 'self-test'])
def test_rccl_link_negotiation_never_strands_a_rank(fail_phase):
    """with p2p_backend 'auto' an ASYMMETRIC failure (one rank cannot draw an id / initialise its communicator / sees a corrupted
    self-test pattern) must end with every rank agreeing on the fallback -- not with its neighbours blocked in a broadcast or a rendezvous."""
    world = 3
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_negotiate_worker, args=(_free_port(), d, world, 1, fail_phase), nprocs=world, join=True)
        res = [torch.load(os.path.join(d, f'n{r}.pt'), weights_only=False) for r in range(world)]
    assert [r['link'] for r in res] == [fail_phase == 'none'] * world
    if fail_phase == 'probe':
        assert all('init' not in r['calls'] for r in res)                      # nobody started connecting
    if fail_phase in ('connect-id', 'connect-init'):
        assert all('selftest' not in r['calls'] for r in res)                  # nobody started the self-test
        assert 'destroy' in res[0]['calls'] or 'destroy' in res[2]['calls'] or fail_phase == 'connect-id'
    if fail_phase == 'connect-id':
        assert 'init' in res[0]['calls'] and res[2]['calls'].count('init') == 0      # pair (0, 1) connected; pair (1, 2) skipped its rendezvous on BOTH ends
