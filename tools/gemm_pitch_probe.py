"""Does the row pitch of an operand matter?  The SDXL step's big GEMMs run tile-independent times (profiles/r3_gemm_desc_ledger.jsonl: [1024, 10240, 1280] NT
58 - 65 us on every tile from 64^2 to 256^2, hipBLASLt 35 us): a candidate cause is L2 / HBM channel aliasing of row pitches that are multiples of 256 B x 2^k
(2 560, 10 240, 20 480 B).  This probe re-times the heaviest descriptors of a saved trace with each operand's pitch padded by `pad` elements (A, B, C alone and
together), HBM-cold (operands rotate through 600 MB) and cache-warm (4 operand sets, < the 256 MiB Infinity Cache), against the unpadded launch.

    python tools/gemm_pitch_probe.py profiles/r2_gemm_trace_sdxl_step.json [out.jsonl] [--top=14] [--pads=0,64,136]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import gemm_desc_timing, gemm_replay  # noqa: E402


def main():
    from diffusion_pipe_amd import ops
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    opts = dict(a[2:].split('=', 1) if '=' in a else (a[2:], '1') for a in sys.argv[1:] if a.startswith('--'))
    uniq = [{k: (tuple(v) if isinstance(v, list) else v) for k, v in d.items()} for d in json.load(open(args[0]))]
    uniq = [d for d in uniq if d['bo'] * d['bi'] == 1 and d['dt'] == 0]
    uniq.sort(key=lambda d: -d['count'] * gemm_replay.flops(d))
    uniq = uniq[:int(opts.get('top', 14))]
    pads = [int(x) for x in opts.get('pads', '0,64,136').split(',')]
    out = open(args[1], 'w') if len(args) > 1 else None
    dev = torch.device('cuda:0')
    arena = gemm_replay.Arena(dev, 3 << 30)
    ops.WS_LANE = 'pitch-probe'
    for d in uniq:
        rec = {'ta': d['ta'], 'tb': d['tb'], 'M': d['M'], 'N': d['N'], 'K': d['K'], 'count': d['count'], 'acc': int(d['acc']), 'lda': d['lda'], 'ldb': d['ldb'], 'ldc': d['ldc'],
               'gflop': round(gemm_replay.flops(d) / 1e9, 2), 'cold_us': {}, 'warm_us': {}}
        for pad in pads:
            for which in (('a',), ('b',), ('c',), ('a', 'b', 'c')) if pad else ((),):
                dd = dict(d)
                for w in which:
                    dd['ld' + w] = d['ld' + w] + pad
                if d['res']:
                    dd['ldr'] = dd['ldc']
                name = f"pad{pad}_{''.join(which)}" if pad else 'plain'
                rec['cold_us'][name] = round(gemm_desc_timing.time_desc(dd, 0, dev, arena, ops), 2)
                rec['warm_us'][name] = round(gemm_desc_timing.time_desc(dd, 0, dev, arena, ops, budget_bytes=1), 2)
        line = json.dumps(rec)
        print(line, flush=True)
        if out:
            out.write(line + '\n'); out.flush()


if __name__ == '__main__':
    main()
