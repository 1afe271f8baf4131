"""In-graph kernel timing on the GPU box: each op is captured N times into one hipGraph and replayed, so the number is
GPU time per launch (no host launch gaps).  Prints JSON lines: the pipelined GEMM vs the generic kernel vs the
PyTorch-ROCm library (hipBLASLt) on the SDXL shapes of one micro-batch, plus the column-sum / LayerNorm-backward kernels."""
import json
import sys

import torch

sys.path.insert(0, '.')
from diffusion_pipe_amd import ops  # noqa: E402


def graph_time(fn, n=20, reps=5):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * n) * 1e3     # us


SHAPES = [  # (ta, tb, M, N, K, count per micro-batch)
    (0, 0, 1024, 1280, 1280, 372), (0, 0, 1024, 1280, 10240, 60), (1, 0, 1280, 1280, 1024, 372), (1, 0, 10240, 1280, 1024, 60),
    (0, 1, 1024, 1280, 1280, 372), (0, 1, 1024, 1280, 5120, 60), (0, 1, 1024, 10240, 1280, 60), (1, 0, 640, 640, 4096, 70),
    (0, 0, 1024, 5120, 1280, 60), (1, 0, 1280, 5120, 1024, 60), (0, 1, 77, 1280, 2048, 120), (0, 0, 77, 1280, 1280, 128),
    (0, 1, 77, 1280, 1280, 128), (0, 0, 77, 2048, 1280, 120), (0, 0, 77, 1280, 5120, 32), (0, 1, 77, 1280, 5120, 32),
    (1, 0, 1280, 1280, 77, 128), (1, 0, 1280, 2048, 77, 120), (0, 0, 4096, 640, 640, 70), (0, 1, 4096, 640, 640, 70),
    (1, 0, 5120, 640, 4096, 10), (0, 1, 4096, 5120, 640, 10), (0, 0, 4096, 640, 5120, 10), (0, 1, 4096, 640, 2560, 10),
    (0, 0, 4096, 2560, 640, 10), (1, 0, 640, 2560, 4096, 10), (0, 1, 77, 768, 768, 48), (0, 1, 1024, 3840, 1280, 0),
    (0, 1, 8192, 8192, 8192, 0), (0, 1, 4608, 9216, 3072, 0)]


def main():
    dev = torch.device('cuda:0')
    tot = {'pipe': 0.0, 'g64': 0.0, 'g128': 0.0, 'torch': 0.0, 'best_generic': 0.0}
    for (ta, tb, M, N, K, cnt) in SHAPES:
        a = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
        b = torch.randn((N, K) if tb else (K, N), device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        aa, bb = (a.t() if ta else a), (b.t() if tb else b)
        rec = {'op': 'gemm', 'ta': ta, 'tb': tb, 'M': M, 'N': N, 'K': K, 'cnt': cnt}
        fl = 2.0 * M * N * K
        for name, hint in (('pipe', 1000), ('t64s1', 2001), ('t64s2', 2002), ('t64s4', 2004), ('t64s8', 2008), ('t128s1', 3001), ('t128s3', 3003),
                           ('g64', 64), ('g128', 128)):
            try:
                us = graph_time(lambda: ops.mm(a, b, bool(ta), bool(tb), out=out, tile_hint=hint))
            except Exception as e:  # not eligible
                us = float('nan')
            rec[name + '_us'] = round(us, 1)
        us = graph_time(lambda: torch.matmul(aa, bb, out=out))
        rec['torch_us'] = round(us, 1)
        rec['pipe_TF'] = round(fl / rec['pipe_us'] / 1e6, 1) if rec['pipe_us'] == rec['pipe_us'] else None
        rec['torch_TF'] = round(fl / us / 1e6, 1)
        for k in ('pipe', 'g64', 'g128', 'torch'):
            v = rec[k + '_us']
            tot[k] += cnt * (v if v == v else min(rec['g64_us'], rec['g128_us'])) / 1e3
        tot['best_generic'] += cnt * min(rec['g64_us'], rec['g128_us']) / 1e3
        print(json.dumps(rec), flush=True)
    print(json.dumps({'summary_ms_per_microbatch': {k: round(v, 2) for k, v in tot.items()}}), flush=True)
    # column sums / LayerNorm backward
    for rows, cols in [(1024, 1280), (4096, 640), (1024, 10240), (77, 1280)]:
        x = torch.randn(rows, cols, device=dev, dtype=torch.bfloat16)
        gy = torch.randn(rows, cols, device=dev, dtype=torch.bfloat16)
        acc = torch.zeros(cols, device=dev, dtype=torch.bfloat16)
        us = graph_time(lambda: ops.column_sum(x, out=acc))
        w = torch.ones(cols, device=dev, dtype=torch.bfloat16, requires_grad=True)
        bz = torch.zeros(cols, device=dev, dtype=torch.bfloat16, requires_grad=True)
        xr = x.clone().requires_grad_(True)

        def ln_fwd_bwd():                       # forward inside the captured region too (backward runs on the forward's stream)
            ops.layer_norm_modulate(xr, w, bz, None, None, 1e-5).backward(gy)
        us_lnf = graph_time(lambda: ops.layer_norm_modulate(x, w, bz, None, None, 1e-5))
        us_ln = graph_time(ln_fwd_bwd) - us_lnf
        print(json.dumps({'op': 'colsum/ln', 'rows': rows, 'cols': cols, 'colsum_us': round(us, 1), 'ln_bwd_us': round(us_ln, 1), 'ln_fwd_us': round(us_lnf, 1)}), flush=True)


def attn_main():
    dev = torch.device('cuda:0')
    F = torch.nn.functional
    for (B, Sq, Sk, H, D, causal, cnt) in [(1, 1024, 1024, 20, 64, 0, 60), (1, 4096, 4096, 10, 64, 0, 10), (1, 1024, 77, 20, 64, 0, 60),
                                           (1, 4096, 77, 10, 64, 0, 10), (1, 77, 77, 20, 64, 1, 32), (1, 77, 77, 12, 64, 1, 12),
                                           # head_dim 128, long sequences: Flux 1024^2 (4096 image + 512 text tokens, 24 heads), Wan-14B 512x512x33f (9216 tokens, 40 heads),
                                           # HunyuanVideo 720p x 65f (61 456 tokens; 2 of its 24 heads as a probe)
                                           (1, 4608, 4608, 24, 128, 0, 0), (1, 9216, 9216, 40, 128, 0, 0), (1, 61456, 61456, 2, 128, 0, 0)]:
        q = torch.randn(B, Sq, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        k = torch.randn(B, Sk, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        v = torch.randn(B, Sk, H, D, device=dev, dtype=torch.bfloat16, requires_grad=True)
        go = torch.randn(B, Sq, H, D, device=dev, dtype=torch.bfloat16)
        with torch.no_grad():
            f_us = graph_time(lambda: ops.attention(q, k, v, impl='flash', causal=bool(causal)), n=10)
            tf_us = graph_time(lambda: F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=bool(causal)), n=10)
        fb_us = graph_time(lambda: ops.attention(q, k, v, impl='flash', causal=bool(causal)).backward(go), n=10)
        tfb_us = graph_time(lambda: F.scaled_dot_product_attention(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), is_causal=bool(causal)).backward(go.transpose(1, 2)), n=10)
        fl = 4.0 * Sq * Sk * D * H * B * (0.5 if causal else 1.0)
        print(json.dumps({'op': 'attn', 'Sq': Sq, 'Sk': Sk, 'H': H, 'D': D, 'causal': causal, 'cnt': cnt, 'fwd_us': round(f_us, 1), 'bwd_us': round(fb_us - f_us, 1),
                          'fwd_TF': round(fl / f_us / 1e6, 1), 'bwd_TF': round(2.5 * fl / max(fb_us - f_us, 1e-9) / 1e6, 1), 'torch_fwd_TF': round(fl / tf_us / 1e6, 1),
                          'torch_fwd_us': round(tf_us, 1), 'torch_bwd_us': round(tfb_us - tf_us, 1),
                          'tot_ms': round(cnt * fb_us / 1e3, 2), 'torch_tot_ms': round(cnt * tfb_us / 1e3, 2)}), flush=True)


def cold_main():
    """GEMMs whose weight (or gradient accumulator) comes cold from HBM, as in a training step: the graph cycles through
    enough distinct weight buffers to exceed the 256 MiB Infinity Cache; activations stay warm."""
    dev = torch.device('cuda:0')
    cases = [(0, 1, 1024, 1280, 1280), (0, 0, 1024, 1280, 1280), (1, 0, 1280, 1280, 1024), (0, 1, 1024, 3840, 1280), (0, 0, 1024, 1280, 3840),
             (1, 0, 3840, 1280, 1024), (0, 1, 1024, 10240, 1280), (1, 0, 10240, 1280, 1024), (0, 0, 1024, 1280, 10240),
             (0, 1, 1024, 1280, 5120), (0, 0, 1024, 5120, 1280), (1, 0, 1280, 5120, 1024), (0, 1, 4096, 640, 2560), (0, 0, 4096, 640, 5120),
             (1, 0, 640, 640, 4096), (1, 0, 640, 1920, 4096), (0, 1, 4096, 5120, 640), (1, 0, 5120, 640, 4096), (0, 1, 4096, 1920, 640),
             (0, 0, 4096, 640, 1920), (0, 0, 4096, 640, 640), (0, 1, 4096, 640, 640)]
    for (ta, tb, M, N, K) in cases:
        wgrad = bool(ta)
        wbytes = (M * N if wgrad else N * K) * 2
        nbuf = max(2, min(256, -(-800_000_000 // wbytes)))
        a = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
        if wgrad:
            b = torch.randn(K, N, device=dev, dtype=torch.bfloat16)
            outs = [torch.zeros(M, N, device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
            ws = [b] * nbuf
        else:
            ws = [torch.randn((N, K) if tb else (K, N), device=dev, dtype=torch.bfloat16) for _ in range(nbuf)]
            outs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16)] * nbuf
        rec = {'op': 'gemm_cold', 'ta': ta, 'tb': tb, 'M': M, 'N': N, 'K': K, 'nbuf': nbuf}
        for name, hint in (('auto', 0), ('t64', 2001), ('t128', 3001), ('t128r2', 4001), ('t128k2', 3002), ('t128k3', 3003), ('t64k2', 2002),
                           ('t128r2k2', 4002), ('t128r2k3', 4003), ('t128s5', 8001), ('t128s5k2', 8002), ('t128s5k3', 8003)):
            def run():
                for i in range(nbuf):
                    ops.mm(a, ws[i], bool(ta), bool(tb), out=outs[i], tile_hint=hint, accumulate=wgrad)
            try:
                rec[name + '_us'] = round(graph_time(run, n=1, reps=3) / nbuf, 1)
            except Exception:       # configuration not eligible for this problem
                rec[name + '_us'] = None
        aa = a.t() if ta else a

        def run_t():
            for i in range(nbuf):
                if wgrad:
                    outs[i].addmm_(aa, ws[i])
                else:
                    torch.matmul(aa, ws[i].t() if tb else ws[i], out=outs[i])
        rec['torch_us'] = round(graph_time(run_t, n=1, reps=3) / nbuf, 1)
        print(json.dumps(rec), flush=True)


def large_main():
    """DiT-sized linears (Flux D=3072 / S=4608, Wan-14B D=5120 / S=9216) and 8192^3: tile configurations vs hipBLASLt, with a
    correctness check of each configuration against the fp32 product of the same bf16 operands."""
    dev = torch.device('cuda:0')
    cases = [(0, 1, 8192, 8192, 8192), (0, 1, 4608, 9216, 3072), (0, 0, 4608, 3072, 9216), (1, 0, 9216, 3072, 4608),
             (0, 1, 4608, 12288, 3072), (0, 1, 4608, 3072, 12288), (0, 1, 9216, 5120, 5120), (0, 0, 9216, 5120, 13824),
             (1, 0, 13824, 5120, 9216), (0, 1, 9216, 13824, 5120), (0, 1, 4096, 4096, 4096), (0, 1, 2048, 2048, 2048)]
    for (ta, tb, M, N, K) in cases:
        a = torch.randn((K, M) if ta else (M, K), device=dev, dtype=torch.bfloat16)
        b = torch.randn((N, K) if tb else (K, N), device=dev, dtype=torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        aa, bb = (a.t() if ta else a), (b.t() if tb else b)
        want = (aa[:256].float() @ bb.float())
        fl = 2.0 * M * N * K
        rec = {'op': 'gemm_large', 'ta': ta, 'tb': tb, 'M': M, 'N': N, 'K': K}
        for name, hint in (('auto', 0), ('t128', 3001), ('t128r2', 4001), ('t256x128', 5001), ('t256', 7001), ('t256k32', 9001)):
            out.zero_()
            ops.mm(a, b, bool(ta), bool(tb), out=out, tile_hint=hint)
            err = ((out[:256].float() - want).abs().max() / want.abs().max()).item()
            us = graph_time(lambda: ops.mm(a, b, bool(ta), bool(tb), out=out, tile_hint=hint), n=5, reps=3)
            rec[name] = [round(us, 1), round(fl / us / 1e6), round(err, 4)]
        us = graph_time(lambda: torch.matmul(aa, bb, out=out), n=5, reps=3)
        rec['torch'] = [round(us, 1), round(fl / us / 1e6)]
        print(json.dumps(rec), flush=True)


if __name__ == '__main__':
    if 'large' in sys.argv[1:]:
        large_main()
    elif 'cold' in sys.argv[1:]:
        cold_main()
    elif 'attn' in sys.argv[1:]:
        attn_main()
    else:
        main()
