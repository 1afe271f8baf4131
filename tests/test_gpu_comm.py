"""Stage-to-stage point-to-point on the C-ABI RCCL wrappers (csrc/comm.hip: dpipe_comm_* / dpipe_send / dpipe_recv, SURVEY.md 8(b) B3;
reference call sites utils/patches.py:126-160 -> DeepSpeed p2p -> NCCL).
  * one GPU: a 1-rank communicator sending to itself inside a group -- the whole wrapper path (run-time RCCL resolution, unique id, communicator,
    grouped send + recv, stream ordering) with real RCCL kernels;
  * >= 2 GPUs (skipped otherwise): the pipeline engine at pp = 2 over `RcclLink` (grouped tuples, receives straight into the stage graphs' static
    buffers) against the single-stage engine."""
import ctypes
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_wrappers_loopback_on_one_gpu(gpu):
    from diffusion_pipe_amd import hip
    lib = hip.lib()
    uid = ctypes.create_string_buffer(128)
    hip.check(lib.dpipe_comm_unique_id(uid), 'unique_id')
    comm = ctypes.c_void_p()
    hip.check(lib.dpipe_comm_init(ctypes.byref(comm), 1, 0, uid.raw), 'comm_init')
    try:
        side = torch.cuda.Stream(gpu)
        src = [torch.randn(1000, 33, device=gpu).to(torch.bfloat16), torch.arange(77, device=gpu, dtype=torch.int64), torch.randn(5, device=gpu)]
        dst = [torch.zeros_like(t) for t in src]
        side.wait_stream(torch.cuda.current_stream(gpu))
        hip.check(lib.dpipe_group_start(), 'group_start')
        for s, d in zip(src, dst):          # one grouped operation: three sends and three receives with the same peer (itself)
            hip.check(lib.dpipe_send(comm, s.data_ptr(), s.numel() * s.element_size(), 0, side.cuda_stream), 'send')
            hip.check(lib.dpipe_recv(comm, d.data_ptr(), d.numel() * d.element_size(), 0, side.cuda_stream), 'recv')
        hip.check(lib.dpipe_group_end(), 'group_end')
        torch.cuda.current_stream(gpu).wait_stream(side)
        torch.cuda.synchronize()
        for s, d in zip(src, dst):
            assert torch.equal(s, d)
        assert lib.dpipe_send(comm, None, 16, 0, side.cuda_stream) < 0 and b'dpipe_send' in lib.dpipe_last_error()      # argument errors are reported, not crashed on
    finally:
        hip.check(lib.dpipe_comm_destroy(comm), 'comm_destroy')


WORKER = r'''
import json, os, sys
sys.path.insert(0, os.environ['DPIPE_ROOT'])
import torch
import torch.distributed as dist
from diffusion_pipe_amd.data import split_batch
from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
from diffusion_pipe_amd.engine.p2p import RcclLink
from diffusion_pipe_amd.workloads import sdxl

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
out_path = sys.argv[1]
torch.cuda.set_device(rank if world > 1 else 0)
dev = torch.device('cuda', rank if world > 1 else 0)
if world > 1:
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
cfg = sdxl.tiny_config()
gas = 4
work = sdxl.SDXLWorkload(cfg, model_config={'min_snr_gamma': 5.0}, dtype=torch.bfloat16, seed=2, device=dev)
module = ManualPipelineModule(layers=work.to_layers(), num_stages=world, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0,
                                                     'hip_graph': True, 'p2p_backend': 'rccl', 'clip_norm_scope': 'global'}, device=dev)
assert world == 1 or isinstance(engine.link, RcclLink)
params = [p for p in module.parameters() if p.requires_grad]
engine._configure_optimizer(lambda ps: torch.optim.SGD(ps, lr=1e-3) if len(ps) else None, params)
res = []
for step in range(3):
    torch.manual_seed(100 + step)
    feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=32, seed=10 + step))
    micro = split_batch((feats, label), gas)
    need = engine.is_first_stage() or engine.is_last_stage()
    engine.reset_activation_shape()
    loss = engine.train_batch(iter(micro) if need else None)
    res.append((loss.item(), engine.get_global_grad_norm().item()))
if rank == 0:
    json.dump({'world': world, 'res': res}, open(out_path, 'w'))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _run(tmp_path, world):
    script = tmp_path / 'worker.py'
    script.write_text(WORKER)
    out = tmp_path / f'rccl_{world}.json'
    env = dict(os.environ, DPIPE_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY='0', DEBUG_CLR_GRAPH_PACKET_CAPTURE='0')
    if world == 1:
        env.update(RANK='0', WORLD_SIZE='1')
        cmd = [sys.executable, str(script), str(out)]
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
               str(script), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return json.loads(out.read_text())


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two MI355X (xGMI peers)')
def test_pp2_over_rccl_link_matches_single_stage_engine(gpu, tmp_path):
    base, pp2 = _run(tmp_path, 1), _run(tmp_path, 2)
    for (l0, n0), (l1, n1) in zip(base['res'], pp2['res']):
        assert abs(l1 - l0) / abs(l0) < 2e-2 and abs(n1 - n0) / n0 < 3e-2, (l0, l1, n0, n1)
