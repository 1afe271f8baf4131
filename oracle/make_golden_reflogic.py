"""ORACLE fixture generator (test infrastructure): runs the REFERENCE'S OWN function bodies for the host logic of the hot path
and records their outputs in tests/golden/reflogic.json / reflogic.safetensors, so oracle/intlogic.py, oracle/eager_step.py and
the product's host code are pinned against the reference itself rather than against a reading of it.

The reference modules cannot be imported here (deepspeed / diffusers / peft are absent), so each function is lifted out of its file
with `ast` AT GENERATION TIME -- nothing of the reference is stored in this repository -- compiled under its own file name, and
executed in a namespace that holds only what the function needs.  Third-party pieces the functions lean on are stubbed and named
as such below ([3P]): DeepSpeed's TrainSchedule helpers and instruction classes, its accelerator / process-group helpers.

Run in the build container:   python oracle/make_golden_reflogic.py
"""
import ast
import hashlib
import json
import math
import os
import random

import numpy as np
import torch
import torch.nn.functional as F
from safetensors.torch import save_file

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'tests', 'golden')


def lift(path, name, cls=None, namespace=None):
    """Compile function `name` (a method of `cls` when given) from reference file `path` and return the function object."""
    full = os.path.join(REF, path)
    tree = ast.parse(open(full).read(), filename=full)
    body = tree.body
    if cls is not None:
        body = next(n for n in body if isinstance(n, ast.ClassDef) and n.name == cls).body
    node = next(n for n in body if isinstance(n, ast.FunctionDef) and n.name == name)
    node.decorator_list = []
    ns = dict(namespace or {})
    exec(compile(ast.Module(body=[node], type_ignores=[]), full, 'exec'), ns)
    return ns[name], f'{path}:{node.lineno}-{node.end_lineno}'


def lift_classes(path, names, namespace):
    """Compile the named top-level classes of a reference file (autocast decorators dropped: CPU fp32 run)."""
    full = os.path.join(REF, path)
    tree = ast.parse(open(full).read(), filename=full)
    nodes = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in names]
    for cls in nodes:
        for fn in cls.body:
            if isinstance(fn, ast.FunctionDef):
                fn.decorator_list = [d for d in fn.decorator_list if 'autocast' not in ast.unparse(d)]
    ns = dict(namespace)
    exec(compile(ast.Module(body=nodes, type_ignores=[]), full, 'exec'), ns)
    return ns


# ------------------------------------------------------------------------------------------------ [3P] stubs
class _Instr:
    def __init__(self, *args):
        self.args = args

    def tag(self):
        return [self.__class__.__name__, self.args[0] if self.args else None]


INSTRUCTIONS = {n: type(n, (_Instr,), {}) for n in ('LoadMicroBatch', 'ForwardPass', 'BackwardPass', 'SendActivation', 'RecvActivation',
                                                    'SendGrad', 'RecvGrad', 'ReduceTiedGrads', 'ReduceGrads', 'OptimizerStep')}


class TrainScheduleStub:
    """[3P] deepspeed.runtime.pipe.schedule.TrainSchedule helper methods (DeepSpeed 0.18.4, restated: not in the snapshot)."""

    def __init__(self, micro_batches, stages, stage_id):
        self.micro_batches, self.stages, self.stage_id = micro_batches, stages, stage_id
        self.prev_stage, self.next_stage = stage_id - 1, stage_id + 1

    def _valid_micro_batch(self, mb):
        return 0 <= mb < self.micro_batches

    def _valid_stage(self, s):
        return 0 <= s < self.stages

    def num_pipe_buffers(self):
        return max(2, min(self.stages - self.stage_id, self.micro_batches))

    def _buffer_idx(self, mb):
        return mb % self.num_pipe_buffers()

    def _step_to_micro_batch(self, step_id):
        even_step, even_stage = step_id % 2 == 0, self.stage_id % 2 == 0
        if even_step and even_stage:
            return step_id // 2 - self.stage_id // 2, True
        if not even_step and not even_stage:
            return (step_id - 1) // 2 - self.stage_id // 2, True
        if even_step and not even_stage:
            return step_id // 2 - self.stages + (self.stage_id + 1) // 2, False
        return (step_id - 1) // 2 - self.stages + 1 + self.stage_id // 2, False


def loader_trace(make_loader):
    """The scenario both loaders are put through: 3 dataset batches x GAS 2; 8 micro-batches, a state dict after the 3rd, a fresh loader
    resumed from it, 7 more micro-batches.  Records (payload id, epoch, num_batches_pulled) after every next()."""
    class DS(torch.utils.data.Dataset):
        def __len__(self):
            return 3

        def __getitem__(self, i):
            return {'x': torch.full((4, 2), float(i)), 'mask': None}
    model = type('M', (), {'prepare_inputs': staticmethod(lambda batch, timestep_quantile=None: ((batch['x'],), (batch['x'] + 100, batch['mask'])))})
    engine = type('E', (), {'is_pipe_parallel': False})()

    def run(loader, n):
        it = iter(loader)
        out = []
        for _ in range(n):
            f, l = next(it)
            out.append([float(f[0][0, 0]), int(f[0].shape[0]), float(l[0][0, 0]), int(l[1].numel()), loader.epoch, loader.num_batches_pulled])
        return out
    a = make_loader(DS(), engine, 2, model)
    first = run(a, 3)
    state = a.state_dict()
    more = run(a, 5)
    b = make_loader(DS(), engine, 2, model)
    b.load_state_dict(dict(state))
    resumed = run(b, 7)
    # a state dict taken exactly at an epoch boundary (what save_checkpoint stores when process_epoch fires): the loader has just rolled
    # over, num_batches_pulled == 0, and load_state_dict turns that into skip = -1 -- the sampler then starts at index -1
    c = make_loader(DS(), engine, 2, model)
    run(c, 6)
    boundary_state = c.state_dict()
    d = make_loader(DS(), engine, 2, model)
    d.load_state_dict(dict(boundary_state))
    boundary_resumed = run(d, 9)
    return {'len': len(a), 'first': first, 'state': state, 'more': more, 'resumed': resumed, 'resumed_state': b.state_dict(),
            'boundary_state': boundary_state, 'boundary_resumed': boundary_resumed}


def main():
    meta, gold, tensors = {}, {}, {}

    # ---- a2: patched TrainSchedule.steps (utils/patches.py:113-160) --------------------------------------------------------
    fn, meta['train_schedule_steps'] = lift('utils/patches.py', 'train_schedule_steps', namespace=INSTRUCTIONS)
    gold['train_schedule'] = {}
    for stages, mbs in [(1, 1), (1, 4), (2, 1), (2, 4), (3, 5), (4, 8), (4, 2), (8, 16), (8, 3)]:
        for stage in range(stages):
            steps = [[c.tag() for c in cmds] for cmds in fn(TrainScheduleStub(mbs, stages, stage))]
            gold['train_schedule'][f'{stages},{mbs},{stage}'] = steps

    # ---- a1: ManualPipelineModule._partition_layers, manual branch (utils/pipeline.py:16-53) --------------------------------
    fn, meta['_partition_layers'] = lift('utils/pipeline.py', '_partition_layers', cls='ManualPipelineModule',
                                         namespace={'LayerSpec': type('LayerSpec', (), {}), 'nn': torch.nn})
    gold['manual_partition'] = []
    for L, split in [(23, [10]), (23, [5, 11, 17]), (59, [29]), (42, [10, 21, 31]), (63, [7, 15, 23, 31, 39, 47, 55]), (6, [2])]:
        for stage in range(len(split) + 1):
            bounds = {}

            class Self:
                manual_partition_split = split
                global_rank = 1                      # not rank 0: skips the listing printout
                _layer_specs = [None] * L
                loss_fn = None

                class _topo:
                    get_dim = staticmethod(lambda axis: len(split) + 1)
                    get_coord = staticmethod(lambda rank: type('C', (), {'pipe': stage}))

                def _set_bounds(self, start, stop):
                    bounds.update(start=start, stop=stop)
            s = Self()
            fn(s, 'manual')
            gold['manual_partition'].append({'layers': L, 'split': split, 'stage': stage, 'parts': s.parts, 'bounds': [bounds['start'], bounds['stop']]})

    # ---- a11: index arithmetic (utils/common.py, utils/dataset.py) ---------------------------------------------------------
    rn, meta['round_to_nearest_multiple'] = lift('utils/common.py', 'round_to_nearest_multiple')
    rd, meta['round_down_to_multiple'] = lift('utils/common.py', 'round_down_to_multiple')
    xs = [0.0, 15.9, 16.0, 48.0, 80.0, 112.0, 1023.5, 1024.0, 1040.0, 1056.0, 777.77, 31.999, 4096.3]
    gold['round_to_nearest_multiple'] = [[x, m, rn(x, m)] for x in xs for m in (8, 16, 32, 64)]
    gold['round_down_to_multiple'] = [[x, m, rd(x, m)] for x in xs for m in (8, 16, 32)]
    ds_ns = {'np': np, 'random': random, 'hashlib': hashlib, 'ROUND_DECIMAL_DIGITS': 3, 'torch': torch}
    dedup, meta['dedup_and_sort'] = lift('utils/dataset.py', 'dedup_and_sort', namespace=ds_ns)
    seed_h, meta['seed_from_hash'] = lift('utils/dataset.py', 'seed_from_hash', namespace=ds_ns)
    shuf, meta['shuffle_with_seed'] = lift('utils/dataset.py', 'shuffle_with_seed', namespace=ds_ns)
    split_b, meta['split_batch'] = lift('utils/dataset.py', 'split_batch', namespace=ds_ns)
    ars_in = list(np.geomspace(0.5, 2.0, num=7)) + [1.0, 1.00049, 0.99951, 2.0]
    gold['dedup_and_sort'] = {'in': [float(a) for a in ars_in], 'out': [float(a) for a in dedup(ars_in)]}
    gold['seed_from_hash'] = [[str(x), seed_h(x)] for x in ['a.png', ('/data/x.jpg', 3), 12345, 'caption with spaces']]
    gold['shuffle_with_seed'] = []
    for n, seed in [(10, 0), (10, 1), (37, 123456789), (5, None)]:
        items = list(range(n))
        random.seed(99)
        shuf(items, seed)
        gold['shuffle_with_seed'].append({'n': n, 'seed': seed, 'out': items if seed is not None else None, 'state_preserved': random.random()})
    g = torch.Generator().manual_seed(0)
    feats = (torch.randn(6, 3, generator=g), torch.arange(12).view(6, 2))
    label = (torch.randn(6, 3, generator=g), None)
    pieces = split_b((feats, label), 3)
    gold['split_batch'] = {'pieces': len(pieces), 'none_becomes_empty': [int(p[1][1].numel()) for p in pieces],
                           'rows': [[int(t.shape[0]) for t in p[0]] for p in pieces]}
    tensors['split_batch.in.f0'], tensors['split_batch.in.f1'], tensors['split_batch.in.l0'] = feats[0], feats[1], label[0]
    for i, p in enumerate(pieces):
        tensors[f'split_batch.out{i}.f0'], tensors[f'split_batch.out{i}.f1'], tensors[f'split_batch.out{i}.l0'] = p[0][0].clone(), p[0][1].clone(), p[1][0].clone()

    far, meta['_find_closest_ar_bucket'] = lift('utils/dataset.py', '_find_closest_ar_bucket', cls='DirectoryDataset', namespace=ds_ns)
    fsz, meta['_find_closest_size_bucket'] = lift('utils/dataset.py', '_find_closest_size_bucket', cls='DirectoryDataset', namespace=ds_ns)
    ars = dedup(np.geomspace(0.5, 2.0, num=7))
    frame_buckets = np.array([1, 33, 65])
    size_buckets = np.array(sorted([(512, 512, 1), (640, 384, 1), (384, 640, 1), (512, 512, 33), (640, 384, 33), (512, 512, 65)], key=lambda b: -b[2]))
    stub = type('S', (), {})()
    stub.ars, stub.log_ars, stub.frame_buckets = ars, np.log(ars), frame_buckets
    stub.size_buckets = size_buckets
    stub.log_ars_size = None
    gold['find_closest_ar_bucket'] = {'ars': [float(a) for a in ars], 'frame_buckets': frame_buckets.tolist(), 'cases': []}
    for ar in (0.4, 0.5, 0.7, 0.75, 1.0, 1.3333, 1.7777, 2.5):
        for frames, is_video in ((1, False), (1, True), (20, True), (33, True), (64, True), (65, True), (200, True)):
            r = far(stub, math.log(ar), frames, is_video)
            gold['find_closest_ar_bucket']['cases'].append([ar, frames, is_video, None if r is None else [float(r[0]), int(r[1])]])
    sstub = type('S', (), {})()
    sstub.size_buckets = size_buckets
    sstub.log_ars = np.log(np.array([w / h for w, h, _ in size_buckets]))
    gold['find_closest_size_bucket'] = {'size_buckets': size_buckets.tolist(), 'cases': []}
    for ar in (0.5, 0.6, 1.0, 1.5, 1.6667, 3.0):
        for frames, is_video in ((1, False), (1, True), (32, True), (33, True), (65, True), (100, True)):
            r = fsz(sstub, math.log(ar), frames, is_video)
            gold['find_closest_size_bucket']['cases'].append([ar, frames, is_video, None if r is None else [int(v) for v in r]])

    # ---- a5: timestep distributions (utils/common.py:124-160) -------------------------------------------------------------
    cm_ns = {'torch': torch, 'math': math}
    gtd, meta['get_t_distribution'] = lift('utils/common.py', 'get_t_distribution', namespace=cm_ns)
    std, meta['slice_t_distribution'] = lift('utils/common.py', 'slice_t_distribution', namespace=cm_ns)
    smp, meta['sample_t'] = lift('utils/common.py', 'sample_t', namespace=cm_ns)
    tsh, meta['time_shift'] = lift('utils/common.py', 'time_shift', namespace=cm_ns)
    glf, meta['get_lin_function'] = lift('utils/common.py', 'get_lin_function', namespace=cm_ns)
    for tag, mc in (('logit_normal', {}), ('logit_normal_s1p3', {'sigmoid_scale': 1.3}), ('uniform', {'timestep_sample_method': 'uniform'})):
        t = gtd(mc)
        tensors[f't_dist.{tag}'] = t
        tensors[f't_dist.{tag}.slice_0p2_0p9'] = std(t, 0.2, 0.9).clone()
        tensors[f't_dist.{tag}.quantiles'] = torch.stack([smp(t, 3, quantile=q) for q in (0.0, 0.1, 0.5, 0.9, 0.9999)])
        torch.manual_seed(17)
        tensors[f't_dist.{tag}.sample_seed17'] = smp(t, 8)
    tt = torch.tensor([0.01, 0.25, 0.5, 0.75, 0.99])
    gold['time_shift'] = {'mu': [glf(y1=0.5, y2=1.15)(n) for n in (256, 1024, 4096)], 'tokens': [256, 1024, 4096]}
    tensors['time_shift.t'] = tt
    for n in (256, 1024, 4096):
        tensors[f'time_shift.out.{n}'] = tsh(glf(y1=0.5, y2=1.15)(n), 1.0, tt)

    # ---- a8: loss functions (models/base.py:418-436, models/sdxl.py:281-355,632-651) ---------------------------------------
    glf_base, meta['base.get_loss_fn'] = lift('models/base.py', 'get_loss_fn', cls='BasePipeline', namespace={'torch': torch, 'F': F})
    g = torch.Generator().manual_seed(3)
    out = torch.randn(2, 4, 6, 8, generator=g)
    tgt = torch.randn(2, 4, 6, 8, generator=g)
    mask = (torch.rand(2, 1, 6, 8, generator=g) > 0.3).float()
    tensors['loss.out'], tensors['loss.target'], tensors['loss.mask'] = out, tgt, mask
    gold['default_loss'] = {}
    for tag, cfg in (('mse', {}), ('huber', {'huber_delta': 0.7}), ('smooth_l1', {'smooth_l1_beta': 0.4})):
        loss_fn = glf_base(type('S', (), {'config': cfg})())
        gold['default_loss'][tag] = {'no_mask': loss_fn(out.clone(), (tgt, torch.tensor([]))).item(),
                                     'mask': loss_fn(out.clone(), (tgt, mask.expand_as(out).clone())).item()}
    sd_ns = {'torch': torch, 'F': F}
    prep, meta['prepare_scheduler_for_custom_training'] = lift('models/sdxl.py', 'prepare_scheduler_for_custom_training', namespace=sd_ns)
    snrw, meta['apply_snr_weight'] = lift('models/sdxl.py', 'apply_snr_weight', namespace=sd_ns)
    debi, meta['apply_debiased_estimation'] = lift('models/sdxl.py', 'apply_debiased_estimation', namespace=sd_ns)
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2       # [3P] DDPMScheduler(scaled_linear) of SDXL
    sched = type('Sched', (), {})()
    sched.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    prep(sched)
    tensors['sdxl.all_snr'] = sched.all_snr
    ts = torch.tensor([0, 1, 17, 250, 500, 998, 999])
    per = torch.rand(7, generator=g) + 0.5
    tensors['sdxl.timesteps'], tensors['sdxl.loss_in'] = ts, per
    for vp in (False, True):
        tensors[f'sdxl.min_snr_gamma5.v{int(vp)}'] = snrw(per.clone(), ts, sched, 5.0, vp)
        tensors[f'sdxl.debiased.v{int(vp)}'] = debi(per.clone(), ts, sched, vp)
    sdxl_loss, meta['sdxl.get_loss_fn'] = lift('models/sdxl.py', 'get_loss_fn', cls='SDXLPipeline',
                                               namespace={'torch': torch, 'F': F, 'apply_snr_weight': snrw, 'apply_debiased_estimation': debi})
    gold['sdxl_loss'] = {}
    for tag, mc in (('plain', {}), ('min_snr5', {'min_snr_gamma': 5.0}), ('debiased', {'debiased_estimation_loss': True}),
                    ('v_pred_min_snr5', {'min_snr_gamma': 5.0, 'v_pred': True})):
        stub = type('S', (), {})()
        stub.min_snr_gamma, stub.debiased_estimation_loss = mc.get('min_snr_gamma'), mc.get('debiased_estimation_loss')
        stub.v_pred, stub.scheduler = mc.get('v_pred', False), sched
        loss_fn = sdxl_loss(stub)
        t2 = torch.tensor([17, 600])
        gold['sdxl_loss'][tag] = {'config': mc, 'no_mask': loss_fn((out.clone(), t2), (tgt, torch.tensor([]))).item(),
                                  'mask': loss_fn((out.clone(), t2), (tgt, mask.clone())).item()}

    # ---- a9: clip_grad_norm_ (utils/patches.py:175-246) ------------------------------------------------------------------------
    class _Acc:
        current_device_name = staticmethod(lambda: 'cpu')
        FloatTensor = staticmethod(lambda v: torch.tensor(v, dtype=torch.float32))
    dist_stub = type('dist', (), {'all_reduce': staticmethod(lambda t, op=None, group=None: None), 'get_world_size': staticmethod(lambda group=None: 1),
                                  'ReduceOp': type('R', (), {'MAX': 0, 'SUM': 1})})
    groups_stub = type('groups', (), {'_get_data_parallel_group': staticmethod(lambda: None)})
    ds_stub = type('deepspeed', (), {'runtime': type('rt', (), {'utils': type('u', (), {'is_model_parallel_parameter': staticmethod(lambda p: False)})})})
    clip, meta['clip_grad_norm_'] = lift('utils/patches.py', 'clip_grad_norm_', namespace={
        'torch': torch, 'inf': math.inf, 'get_accelerator': lambda: _Acc, 'dist': dist_stub, 'groups': groups_stub, 'deepspeed': ds_stub})
    g = torch.Generator().manual_seed(11)
    shapes = [(7,), (5, 3), (2, 3, 4), (1,)]
    gold['clip_grad_norm'] = []
    for case, (max_norm, mp_rank) in enumerate([(1.0, None), (0.05, None), (100.0, None), (0.5, 0), (0.5, 1)]):
        params = []
        for i, s in enumerate(shapes):
            p = torch.nn.Parameter(torch.zeros(s))
            p.grad = torch.randn(s, generator=g) * (0.3 + i)
            tensors[f'clip.{case}.grad_in.{i}'] = p.grad.clone()
            params.append(p)
        mpu = None
        if mp_rank is not None:         # pipeline stage `mp_rank`: only model-parallel rank 0 contributes (the hazard of SURVEY 8(a9))
            mpu = type('mpu', (), {'get_model_parallel_rank': staticmethod(lambda r=mp_rank: r), 'get_model_parallel_group': staticmethod(lambda: None)})
        total = clip(params, max_norm, mpu=mpu)
        gold['clip_grad_norm'].append({'max_norm': max_norm, 'mp_rank': mp_rank, 'total_norm': float(total)})
        for i, p in enumerate(params):
            tensors[f'clip.{case}.grad_out.{i}'] = p.grad.clone()

    # ---- a5: prepare_inputs of the three adapters (RNG draw order, quantile path, mask resize, patchify, ids) -----------------------
    from einops import rearrange
    g = torch.Generator().manual_seed(21)
    gold['prepare_inputs'] = {}

    def record(tag, feats, label):
        for i, t in enumerate(feats):
            if torch.is_tensor(t):
                tensors[f'prep.{tag}.f{i}'] = t.clone().contiguous()
        for i, t in enumerate(label):
            if torch.is_tensor(t):
                tensors[f'prep.{tag}.l{i}'] = t.clone().contiguous()
        gold['prepare_inputs'][tag] = {'features': [None if t is None else list(t.shape) for t in feats],
                                       'label': [None if t is None else list(t.shape) for t in label]}

    # SDXL (models/sdxl.py:538-579).  [3P] stubs: tokenizer lookup (ids passed through), DDPMScheduler.add_noise / get_velocity,
    # StableDiffusionXLPipeline._get_add_time_ids
    sdxl_prep, meta['sdxl.prepare_inputs'] = lift('models/sdxl.py', 'prepare_inputs', cls='SDXLPipeline', namespace={'torch': torch, 'F': F})
    acp = torch.cumprod(1.0 - betas, dim=0)

    class _Sched:
        config = type('C', (), {'num_train_timesteps': 1000})

        @staticmethod
        def add_noise(x, n, t):
            a = acp[t].view(-1, 1, 1, 1)
            return a.sqrt() * x + (1 - a).sqrt() * n

        @staticmethod
        def get_velocity(x, n, t):
            a = acp[t].view(-1, 1, 1, 1)
            return a.sqrt() * n - (1 - a).sqrt() * x
    lat4 = torch.randn(3, 4, 16, 24, generator=g)
    msk = (torch.rand(3, 128, 192, generator=g) > 0.4).float()
    ids1, ids2 = torch.randint(1000, 40000, (3, 75), generator=g), torch.randint(1000, 40000, (3, 75), generator=g)
    tensors['prep.sdxl.latents'], tensors['prep.sdxl.mask'], tensors['prep.sdxl.ids1'], tensors['prep.sdxl.ids2'] = lat4, msk, ids1, ids2
    for tag, v_pred, use_mask, q in (('sdxl_eps', False, False, None), ('sdxl_v_mask', True, True, None), ('sdxl_q0p3', False, True, 0.3)):
        st = type('S', (), {})()
        st.scheduler, st.v_pred, st.vae_scale_factor = _Sched, v_pred, 8
        st.tokenizer, st.tokenizer_2 = 'tok1', 'tok2'
        st._get_input_ids = lambda caption, tok: ids1 if tok == 'tok1' else ids2
        st.text_encoder_2 = type('TE', (), {'config': type('C', (), {'projection_dim': 1280})})
        st._get_add_time_ids = lambda orig, crop, tgt, dtype, text_encoder_projection_dim: torch.tensor([list(orig + crop + tgt)], dtype=dtype)
        torch.manual_seed(77)
        feats, label = sdxl_prep(st, {'latents': lat4, 'caption': None, 'mask': msk if use_mask else None}, timestep_quantile=q)
        record(tag, feats, label)

    # Wan (models/wan/wan.py:332-375) over the utils/common.py helpers lifted above
    wan_prep, meta['wan.prepare_inputs'] = lift('models/wan/wan.py', 'prepare_inputs', cls='WanPipeline', namespace={
        'torch': torch, 'F': F, 'get_lin_function': glf, 'time_shift': tsh, 'slice_t_distribution': std, 'sample_t': smp})
    lat5 = torch.randn(2, 16, 3, 12, 16, generator=g)
    msk5 = (torch.rand(2, 96, 128, generator=g) > 0.5).float()
    temb = torch.randn(2, 20, 32, generator=g)
    tensors['prep.wan.latents'], tensors['prep.wan.mask'], tensors['prep.wan.text'] = lat5, msk5, temb
    for tag, mc, use_mask, q in (('wan_plain', {}, False, None), ('wan_shift3_mask', {'shift': 3.0}, True, None),
                                 ('wan_minmax_q', {'min_t': 0.1, 'max_t': 0.8}, False, 0.6), ('wan_flux_shift', {'flux_shift': True}, False, None)):
        st = type('S', (), {})()
        st.model_type, st.cache_text_embeddings, st.model_config, st.t_dist = 't2v', True, mc, gtd(mc)
        torch.manual_seed(78)
        feats, label = wan_prep(st, {'latents': lat5, 'mask': msk5 if use_mask else None, 'text_embeddings': temb,
                                     'seq_lens': torch.tensor([17, 20])}, timestep_quantile=q)
        record(tag, feats, label)

    # Flux (models/flux.py:323-394).  [3P] stub: FluxPipeline._prepare_latent_image_ids
    def _ids(bs, h, w, device, dtype):
        ids = torch.zeros(h, w, 3)
        ids[..., 1] = ids[..., 1] + torch.arange(h)[:, None]
        ids[..., 2] = ids[..., 2] + torch.arange(w)[None, :]
        return ids.reshape(h * w, 3).to(device=device, dtype=dtype)
    flux_tsh, _ = lift('models/flux.py', 'time_shift', namespace={'math': math, 'torch': torch})
    flux_glf, _ = lift('models/flux.py', 'get_lin_function')
    flux_prep, meta['flux.prepare_inputs'] = lift('models/flux.py', 'prepare_inputs', cls='FluxPipeline', namespace={
        'torch': torch, 'F': F, 'rearrange': rearrange, 'get_lin_function': flux_glf, 'time_shift': flux_tsh})
    lat16 = torch.randn(2, 16, 8, 12, generator=g)
    mskf = (torch.rand(2, 64, 96, generator=g) > 0.5).float()
    t5, clip = torch.randn(2, 10, 24, generator=g), torch.randn(2, 12, generator=g)
    tensors['prep.flux.latents'], tensors['prep.flux.mask'], tensors['prep.flux.t5'], tensors['prep.flux.clip'] = lat16, mskf, t5, clip
    for tag, mc, use_mask, q in (('flux_plain', {'guidance': 1.0}, False, None), ('flux_shift_mask', {'guidance': 3.5, 'shift': 3.0}, True, None),
                                 ('flux_fluxshift_uniform', {'guidance': 1.0, 'flux_shift': True, 'timestep_sample_method': 'uniform'}, False, None),
                                 ('flux_q', {'guidance': 1.0, 'sigmoid_scale': 1.3}, False, 0.7)):
        st = type('S', (), {})()
        st.model_config, st.is_flex2, st._prepare_latent_image_ids = mc, False, _ids
        torch.manual_seed(79)
        feats, label = flux_prep(st, {'latents': lat16, 'clip_embed': clip, 't5_embed': t5, 'mask': mskf if use_mask else None}, timestep_quantile=q)
        record(tag, feats, label)

    # ---- a3: PipelineDataLoader (utils/dataset.py:1302-1435): micro-batch stream, epoch roll-over, resume from a state dict ------------
    ld_ns = lift_classes('utils/dataset.py', {'PipelineDataLoader', 'SkipFirstNSampler'}, {'torch': torch, 'split_batch': split_b, 'dist': None})
    gold['loader_trace'] = loader_trace(lambda ds, eng, gas, model: ld_ns['PipelineDataLoader'](ds, eng, gas, model, num_dataloader_workers=0))
    meta['PipelineDataLoader'] = 'utils/dataset.py:1302-1435'

    # ---- a10: get_optimizer (train.py:650-815, nested in the main block) over SDXLPipeline.get_param_groups (models/sdxl.py:604-630) -------
    import inspect
    full = os.path.join(REF, 'train.py')
    tree = ast.parse(open(full).read(), filename=full)
    node = next(n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef) and n.name == 'get_optimizer')
    dummy_cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'DummyOptimizer')
    meta['get_optimizer'] = f'train.py:{node.lineno}-{node.end_lineno}'
    gpg, meta['sdxl.get_param_groups'] = lift('models/sdxl.py', 'get_param_groups', cls='SDXLPipeline', namespace={'is_main_process': lambda: False})
    from collections import defaultdict

    def make_params():
        specs = [('unet.conv_in.weight', (8, 4, 3, 3)), ('unet.conv_in.bias', (8,)), ('unet.mid.attn.to_q.weight', (8, 8)), ('unet.norm.weight', (8,)),
                 ('text_encoder.layers.0.q_proj.weight', (4, 4)), ('text_encoder.layers.0.q_proj.bias', (4,)), ('text_encoder.final_layer_norm.weight', (4,)),
                 ('text_encoder_2.text_projection.weight', (6, 4)), ('text_encoder_2.embeddings.position_embedding.weight', (7, 4)), ('text_encoder_2.ln.bias', (4,))]
        out = []
        for name, shape in specs:
            prm = torch.nn.Parameter(torch.zeros(shape))
            prm.original_name = name
            out.append(prm)
        return out
    gold['get_optimizer'] = []
    for case, (optim_cfg, model_cfg, gbs) in enumerate([
            ({'type': 'adamw', 'lr': 2e-5, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}, {}, 4),
            ({'type': 'AdamW', 'lr': 2e-5, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8, 'beta2_half_life': 2000},
             {'unet_lr': 4e-5, 'text_encoder_2_lr': 1e-6}, 16),
            ({'type': 'sgd', 'lr': 1e-3, 'momentum': 0.9, 'weight_decay': 0.1}, {'text_encoder_1_lr': 5e-4}, 8)]):
        config = {'optimizer': json.loads(json.dumps(optim_cfg))}
        model_stub = type('M', (), {})()
        model_stub.config, model_stub.model_config = config, model_cfg
        model_stub.get_param_groups = lambda params, m=model_stub: gpg(m, params)
        ns = {'torch': torch, 'inspect': inspect, 'config': config, 'global_batch_size': gbs, 'model': model_stub, 'pipeline_model': None,
              'ds_config': {'gradient_accumulation_steps': 1}, 'defaultdict': defaultdict, 'print': lambda *a, **k: None}
        exec(compile(ast.Module(body=[dummy_cls, node], type_ignores=[]), full, 'exec'), ns)
        params = make_params()
        opt = ns['get_optimizer'](params)
        groups = [{'params': [q.original_name for q in gr['params']], **{k: (list(v) if isinstance(v, tuple) else v) for k, v in gr.items()
                                                                          if k in ('lr', 'weight_decay', 'betas', 'eps', 'momentum')}} for gr in opt.param_groups]
        empty = ns['get_optimizer']([])
        gold['get_optimizer'].append({'optimizer': optim_cfg, 'model_config': model_cfg, 'global_batch_size': gbs, 'class': type(opt).__name__,
                                      'groups': groups, 'empty_class': type(empty).__name__, 'empty_groups': empty.param_groups,
                                      'empty_state_dict': empty.state_dict()})

    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, 'reflogic.json'), 'w') as fh:
        json.dump({'generated_from': meta, 'torch': torch.__version__, 'golden': gold}, fh, indent=0)
    save_file({k: v.contiguous() for k, v in tensors.items()}, os.path.join(OUT, 'reflogic.safetensors'))
    print(json.dumps(meta, indent=1))
    print({k: (len(v) if hasattr(v, '__len__') else v) for k, v in gold.items()}, len(tensors), 'tensors')


if __name__ == '__main__':
    main()
