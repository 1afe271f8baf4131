"""The progress-mark primitive (include/dpipe_hip.h C5) inside a PyTorch hipGraph capture, i.e. under PyTorch-ROCm's own HIP runtime: the graph is
   spin(T) -> tanh -> [backward: autograd hook -> dpipe_mark_post] -> spin(T);   after every replay stream B runs dpipe_mark_wait(value = that replay's number) and stamps
an event.  Want per replay: t_mark ~ T (released by THIS replay's mark: not ~0 = an earlier replay's value, not ~2T = the end of the graph), t_graph ~ 2T.
Two replays are also queued back to back before one wait for the SECOND (the engine: a lane replays twice per step, the communication stream waits for the last).
    python tools/mark_capture_probe.py [out.json]"""
import ctypes
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_pipe_amd import hip

dev = torch.device('cuda:0')
lib = hip.lib()


def spin_cycles_for(ms):
    torch.cuda._sleep(1_000_000)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); torch.cuda._sleep(200_000_000); b.record(); torch.cuda.synchronize()
    return int(200_000_000 * ms / a.elapsed_time(b))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    T = 20.0
    cyc = spin_cycles_for(T)
    report = {'torch': torch.__version__, 'hip': torch.version.hip, 'spin_ms': T, 'modes': {}}
    # the textbook primitive first: an external event recorded under capture (an event-record node)
    try:
        ev = torch.cuda.Event(external=True)
        g0 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g0, capture_error_mode='thread_local'):
            torch.cuda._sleep(1000)
            ev.record()
        report['torch_external_event_under_capture'] = 'ok'
    except Exception as e:                              # noqa: BLE001
        report['torch_external_event_under_capture'] = repr(e)[:200]
    torch.cuda.synchronize()
    print('torch.cuda.Event(external=True) under capture:', report['torch_external_event_under_capture'], flush=True)
    for mode in ('thread_local', 'global'):
        w = torch.randn(64, 256, device=dev, requires_grad=True)
        x = torch.randn(64, 256, device=dev)
        mark = torch.zeros(1, dtype=torch.int32, device=dev)
        gen = torch.zeros(1, dtype=torch.int32, device=dev)
        err = torch.zeros(1, dtype=torch.int32).pin_memory()
        A, B = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        posted = []

        def post(_g):
            st = torch.cuda.current_stream(dev)
            hip.check(lib.dpipe_mark_post(mark.data_ptr(), gen.data_ptr(), st.cuda_stream), 'mark_post')
            posted.append(torch.cuda.is_current_stream_capturing())
            torch.cuda._sleep(cyc)                    # the "backward of the earlier layers"
            return None
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(A):
            with torch.cuda.graph(g, stream=A, capture_error_mode=mode):
                torch.cuda._sleep(cyc)
                h = torch.tanh(x * w)
                h.register_hook(post)
                (h * h).sum().backward()
        torch.cuda.synchronize()
        rows, n = [], 0
        for rep in range(5):
            double = rep >= 3                         # two replays queued, one wait for the second
            t0, tm, tg = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(A):
                t0.record(A)
                for _ in range(2 if double else 1):
                    n += 1
                    gen.fill_(n)
                    g.replay()
                tg.record(A)
            with torch.cuda.stream(B):
                hip.check(lib.dpipe_mark_wait(mark.data_ptr(), n, err.data_ptr(), 5000, B.cuda_stream), 'mark_wait')
                tm.record(B)
            torch.cuda.synchronize()
            want_mark = 3 * T if double else T
            want_graph = 4 * T if double else 2 * T
            rows.append({'replay': rep, 'replays_queued': 2 if double else 1, 't_mark_ms': round(t0.elapsed_time(tm), 2), 'want_mark_ms': want_mark,
                         't_graph_ms': round(t0.elapsed_time(tg), 2), 'want_graph_ms': want_graph, 'err': int(err[0])})
            print(mode, rows[-1], flush=True)
        # (replay 0 of a process carries the graph's first-launch cost ahead of its first kernel: judged from replay 1 on)
        ok = all(abs(r['t_mark_ms'] - r['want_mark_ms']) < 0.25 * T and abs(r['t_graph_ms'] - r['want_graph_ms']) < 0.25 * T and r['err'] == 0 for r in rows[1:])
        # a mark that never comes: the wait gives up after its timeout and raises the error word
        t0, tm = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(B):
            t0.record(B)
            hip.check(lib.dpipe_mark_wait(mark.data_ptr(), n + 1000, err.data_ptr(), 50, B.cuda_stream), 'mark_wait')
            tm.record(B)
        torch.cuda.synchronize()
        report['modes'][mode] = {'posted_under_capture': posted, 'replays': rows, 'mark_orders_the_other_stream': ok,
                                 'lost_mark': {'timeout_ms': 50, 'waited_ms': round(t0.elapsed_time(tm), 2), 'err_word': int(err[0])}}
        print(mode, report['modes'][mode]['lost_mark'], 'ok' if ok else 'NOT OK', flush=True)
    print(json.dumps(report))
    if out:
        json.dump(report, open(out, 'w'), indent=1)


if __name__ == '__main__':
    main()
