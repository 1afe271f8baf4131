"""Re-issue a recorded attention call list (ops.ATTN_TRACE: (B, Sq, Sk, H, D, causal, has_backward) per call) on scratch operands and time it with
HIP events: every unique shape is captured into a small hipGraph (forward alone, and forward + backward), its time per call is multiplied by the number
of calls of that shape in the step.  bench.py uses it for the `roofline` object of workloads whose dominant kernel is flash attention (HunyuanVideo:
77 % of the step's FLOPs).  FLOPs: 4 Sq Sk D per head forward (x 0.5 causal), backward counted 2.5 x forward (dQ, dK, dV = 5 products of the same size)."""
from collections import OrderedDict

import torch


def flops_fwd(e):
    B, Sq, Sk, H, D, causal = e[:6]
    return 4.0 * B * Sq * Sk * H * D * (0.5 if causal else 1.0)


def time_list(trace, device, reps=3):
    """-> dict(ms for the whole list, calls, flops, per_shape=[...]).  Every launch of the list is accounted at the measured time of its shape."""
    from diffusion_pipe_amd import ops
    uniq = OrderedDict()
    for e in trace:
        e = tuple(e) + ((1,) if len(e) == 6 else ())
        uniq[e] = uniq.get(e, 0) + 1
    total_ms, total_fl, rows = 0.0, 0.0, []
    for e, cnt in uniq.items():
        B, Sq, Sk, H, D, causal, has_bwd = e
        q = torch.randn(B, Sq, H, D, device=device).to(torch.bfloat16).requires_grad_(bool(has_bwd))
        k = torch.randn(B, Sk, H, D, device=device).to(torch.bfloat16).requires_grad_(bool(has_bwd))
        v = torch.randn(B, Sk, H, D, device=device).to(torch.bfloat16).requires_grad_(bool(has_bwd))
        go = torch.randn(B, Sq, H, D, device=device).to(torch.bfloat16)

        def call():
            if has_bwd:
                ops.attention(q, k, v, impl='flash', causal=bool(causal)).backward(go)
            else:
                with torch.no_grad():
                    ops.attention(q, k, v, impl='flash', causal=bool(causal))
        n = 1 if flops_fwd(e) > 2e12 else 5
        saved, ops.ATTN_TRACE = ops.ATTN_TRACE, None
        try:
            side = torch.cuda.Stream(device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                call()
            torch.cuda.current_stream(device).wait_stream(side)
            torch.cuda.synchronize(device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode='thread_local'):
                for _ in range(n):
                    call()
            g.replay()
            torch.cuda.synchronize(device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize(device)
        finally:
            ops.ATTN_TRACE = saved
        ms = e0.elapsed_time(e1) / (reps * n)
        fl = flops_fwd(e) * (3.5 if has_bwd else 1.0)
        total_ms += ms * cnt
        total_fl += fl * cnt
        rows.append({'shape': list(e), 'calls': cnt, 'ms_per_call': round(ms, 4), 'tflops': round(fl / ms / 1e9, 1)})
        del q, k, v, go, g
    return {'ms': total_ms, 'calls': sum(uniq.values()), 'flops': total_fl, 'per_shape': rows}
