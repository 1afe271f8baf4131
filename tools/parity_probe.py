"""Where does the timed path's gradient differ from the oracle's?  Reproduces bench.py's `parity` leg (SDXL full size, 4 lanes, GAS 8, fused AdamW, W training steps
first), then compares the PRE-clip gradient of every parameter -- [sum |g|, sum g, <g, r>, ||g||_2] rows (oracle/checksums.py) -- between the GPU engine and the
oracle's fp32 eager path on the same weights and micro-batch, and prints the contribution of each module family to the difference of the squared global norm.
Test infrastructure (imports oracle/).   python tools/parity_probe.py [train_steps] [out.json]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class Rows(torch.optim.Optimizer):
    def __init__(self, params, names):
        super().__init__(params, {})
        self.names, self.rows = names, {}

    def step(self, closure=None):
        from oracle.checksums import checksum4
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is not None:
                    self.rows[self.names[id(p)]] = checksum4(p.grad, self.names[id(p)])


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    out_path = sys.argv[2] if len(sys.argv) > 2 else ''
    from diffusion_pipe_amd import optim
    from diffusion_pipe_amd.data import split_batch
    from diffusion_pipe_amd.engine import ManualPipelineModule, initialize
    from diffusion_pipe_amd.workloads import sdxl
    from oracle.cpu_baseline import sdxl_cpu_baseline
    from oracle.checksums import relative_errors
    dev = torch.device('cuda', 0)
    gas, lanes = 8, 4
    cfg = sdxl.SDXLConfig()
    work = sdxl.SDXLWorkload(cfg, dtype=torch.bfloat16, seed=0, device=dev)
    module = ManualPipelineModule(layers=work.to_layers(), num_stages=1, partition_method='parameters', loss_fn=work.get_loss_fn(), dynamic_shape=True)
    engine, _, _, _ = initialize(model=module, config={'train_micro_batch_size_per_gpu': 1, 'gradient_accumulation_steps': gas, 'gradient_clipping': 1.0, 'steps_per_print': 1 << 30,
                                                         'hip_graph': True, 'graph_lanes': lanes}, device=dev)
    params = [p for p in module.parameters() if p.requires_grad]
    work.train_config = {'optimizer': {'type': 'adamw', 'lr': 1e-5, 'betas': [0.9, 0.99], 'weight_decay': 0.01, 'eps': 1e-8}}
    fused = optim.make_optimizer_factory(work.train_config, work, global_batch_size=gas)
    engine._configure_optimizer(fused, params)
    torch.manual_seed(1234)
    pool = []
    for s in range(3):
        feats, label = work.prepare_inputs(sdxl.synthetic_batch(cfg, batch_size=gas, latent_hw=128, seed=100 + s))
        pool.append(split_batch((feats, label), gas))
    sample = pool[0][0]
    for i in range(steps):
        engine.reset_activation_shape()
        engine.train_batch(iter(pool[i % len(pool)]))
    torch.cuda.synchronize()
    if os.environ.get('PROBE_EAGER', '0') == '1':
        # bench.py's roofline leg: ONE more optimizer step, eager (graphs off), with the GEMM launch list recorded; PROBE_ROOFLINE=1 also replays that list as bench.py does
        from diffusion_pipe_amd import ops
        ops.GEMM_TRACE = []
        was = engine.use_graph
        engine.use_graph = False
        for p_ in module.parameters():
            p_.grad = None
        engine.reset_activation_shape()
        engine.train_batch(iter(pool[0]))
        engine.use_graph = was
        trace, ops.GEMM_TRACE = ops.GEMM_TRACE, None
        torch.cuda.synchronize()
        if os.environ.get('PROBE_ROOFLINE', '0') == '1':
            from tools import gemm_replay
            gemm_replay.time_in_graph(trace, dev, reps=3)
            per_lane = trace[:len(trace) // gas * max(1, gas // lanes)]
            gemm_replay.time_concurrent(per_lane, dev, lanes)
            torch.cuda.synchronize()
    names = {id(p): f'{k}.{n}' for k, m in work.modules().items() for n, p in m.named_parameters()}
    if os.environ.get('PROBE_FUSED', '0') == '1':
        # bench.py's own path: the fused step end reads the lanes' bf16 accumulators (sum over lanes in fp32 inside adamw_sumsq / adamw_step).  Record the rows of the
        # fp32 lane sum right before the fused update consumes (and zeroes) the accumulators; the update itself is skipped (weights stay what the oracle gets)
        from tools.parity_report import record_fused_rows
        seen = record_fused_rows(engine, names)
        engine.reset_activation_shape()
        loss = engine.train_batch(iter([sample] * gas)).item()
        norm = engine.get_global_grad_norm().item()
        torch.cuda.synchronize()
        gpu = seen
        print(f'fused step end: adamw_sumsq norm {norm:.6f}; sqrt(sum of per-parameter ||fp32 lane sum||^2) {sum(r[3] ** 2 for r in seen.values()) ** 0.5:.6f}')
    else:
        rec = Rows(params, names)
        engine.optimizer = rec
        engine.reset_activation_shape()
        loss = engine.train_batch(iter([sample] * gas)).item()
        norm = engine.get_global_grad_norm().item()
        torch.cuda.synchronize()
        coef = min(1.0, 1.0 / (norm + 1e-6))
        gpu = {k: [v / coef for v in r] for k, r in rec.rows.items()}
    state = {k: {n: v.detach().to('cpu', torch.float32) for n, v in m.state_dict().items()} for k, m in work.modules().items()}
    cpu = sdxl_cpu_baseline(cfg, latent_hw=128, micro_batch=sample, state=state, per_parameter=True)
    ref = cpu['rows']
    print(f'after {steps} training steps: loss gpu {loss:.7f} cpu {cpu["loss"]:.7f}; grad norm gpu {norm:.6f} cpu {cpu["grad_norm"]:.6f} rel {abs(norm - cpu["grad_norm"]) / cpu["grad_norm"]:.3e}')
    from tools.parity_report import family_table
    table, top = family_table(gpu, ref)
    if out_path:
        json.dump({'train_steps': steps, 'loss_gpu': loss, 'loss_cpu': cpu['loss'], 'grad_norm_gpu': norm, 'grad_norm_cpu': cpu['grad_norm'], 'families': table, 'top': top},
                  open(out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
