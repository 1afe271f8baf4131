"""Is the GEMM list memory-bound under lanes?  The step's recorded launch list (one lane's share) replayed (a) as one graph, (b) as four concurrent lane graphs, with
operand arenas of 3 GiB (HBM-cold, as in training), 512 MiB and 160 MiB (everything stays in the 256 MiB Infinity Cache: no HBM traffic; the largest single problem of the
list needs ~120 MiB, so nothing smaller can hold it).  If the four-lane time barely moves when the operands stop coming from HBM, HBM bandwidth is not what the lanes contend for.

    python tools/gemm_mem_sensitivity.py profiles/r4_gemm_trace_sdxl_step.json [out.json]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gemm_replay  # noqa: E402


def main():
    uniq = json.load(open(sys.argv[1]))
    trace = []
    for d in uniq:                       # one micro-batch's worth of the step (counts are per step of 8 micro-batches)
        d = {k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()}
        n = max(1, d.pop('count') // 8)
        trace += [d] * n
    dev = torch.device('cuda:0')
    out = {'problems_per_lane': len(trace), 'tflop_per_lane': round(sum(gemm_replay.flops(d) for d in trace) / 1e12, 2), 'rows': []}
    for name, nbytes in (('3 GiB (HBM-cold)', 3 << 30), ('512 MiB (twice the Infinity Cache)', 512 << 20), ('160 MiB (Infinity-Cache resident)', 160 << 20)):
        one = gemm_replay.time_in_graph(trace, dev, reps=3, arena_bytes=nbytes)
        four = gemm_replay.time_concurrent(trace, dev, 4, reps=2, arena_bytes=nbytes)
        row = {'arena': name, 'single_stream_ms': round(one['ms'], 2), 'single_stream_TFLOPs': round(one['flops'] / one['ms'] / 1e9, 1),
               'four_lanes_ms': round(four['ms'], 2), 'four_lanes_TFLOPs': round(four['flops'] / four['ms'] / 1e9, 1)}
        out['rows'].append(row)
        print(json.dumps(row), flush=True)
        torch.cuda.empty_cache()
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], 'w'), indent=1)


if __name__ == '__main__':
    main()
