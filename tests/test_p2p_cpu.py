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
