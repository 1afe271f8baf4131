"""Read-side rate of the latent / text-embedding cache: items / s and MB / s of
  (a) the reference's reader (utils/cache.py:24-36: seek + read + BytesIO + torch.load per item) -- only where /root/reference exists (this container),
  (b) this repo's mmap reader (cache.Cache.__getitem__), (c) cache.CachePrefetcher (2 worker threads reading ahead, pinned staging where a GPU exists)
over the SAME cache directory (written once, by the reference's writer when it is importable: the two writers are byte-identical, tests/test_cache_cpu.py), in a
shuffled order, on the host's cores.  What the step needs: 22.6 SDXL items / s, 3.1 Flux items / s, 0.94 Wan items / s (DESIGN.md status).

    python tools/cache_feed_rate.py [out.json]"""
import importlib.util
import json
import os
import random
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference/utils/cache.py'


def items_for(kind, n):
    g = torch.Generator().manual_seed(7)
    for i in range(n):
        if kind == 'sdxl':       # models/sdxl.py:541-589: VAE latents + two token-id rows + the crop / size conditioning scalars
            yield {'latents': torch.randn(4, 128, 128, generator=g), 'input_ids': torch.randint(1000, 40000, (75,), generator=g),
                   'input_ids_2': torch.randint(1000, 40000, (75,), generator=g), 'original_size': (1024, 1024), 'crop': (0, 0), 'caption': f'item {i}'}
        else:                    # models/wan/wan.py:331-375: video latents + umT5 states
            yield {'latents': torch.randn(16, 9, 64, 64, generator=g), 'text_embeddings': torch.randn(512, 4096, generator=g).to(torch.bfloat16), 'seq_lens': 512}


def timed(read, order):
    t0 = time.perf_counter()
    nbytes = 0
    for i in order:
        item = read(i)
        nbytes += sum(v.numel() * v.element_size() for v in item.values() if torch.is_tensor(v))
    dt = time.perf_counter() - t0
    return {'items_per_s': round(len(order) / dt, 1), 'mb_per_s': round(nbytes / dt / 1e6, 1), 'seconds': round(dt, 3)}


def main():
    from diffusion_pipe_amd.cache import Cache, CachePrefetcher
    ref_mod = None
    if os.path.isfile(REF):
        spec = importlib.util.spec_from_file_location('ref_cache', REF)
        ref_mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref_mod)
        import sqlite3
        import types
        # the reference asks for sqlite3.connect(..., autocommit=False) (python >= 3.12); python 3.10's default isolation level opens the same implicit transactions
        ref_mod.sqlite3 = types.SimpleNamespace(connect=lambda path, autocommit=False, **kw: sqlite3.connect(path, **kw))
    out = {'host_cores': os.cpu_count(), 'torch_threads': torch.get_num_threads(), 'writer': 'reference' if ref_mod else 'this repo (byte-identical)'}
    for kind, n in (('sdxl', 512), ('wan', 96)):
        with tempfile.TemporaryDirectory() as d:
            writer = (ref_mod.Cache if ref_mod else Cache)(d, 'fp', shard_size_gb=0.1)
            for item in items_for(kind, n):
                writer.add(item)
            writer.finalize_current_shard()
            writer.con.close() if ref_mod else writer.close()
            order = list(range(n))
            random.Random(3).shuffle(order)
            row = {'items': n, 'shards': len([f for f in os.listdir(d) if f.endswith('.bin')])}
            for rep in range(2):             # second pass = page cache warm for every reader
                if ref_mod:
                    r = ref_mod.Cache(d, 'fp')
                    row['reference_reader'] = timed(lambda i: r[i], order)
                    for f in r.open_files.values():
                        f.close()
                    r.con.close()
                c = Cache(d, 'fp')
                row['mmap_reader'] = timed(lambda i: c[i], order)
                pf = CachePrefetcher(c, order, depth=8, pin=torch.cuda.is_available())
                it = iter(pf)
                row['prefetcher'] = dict(timed(lambda i: next(it), order), workers=pf.workers)
                pf.close()
                c.close()
            out[kind] = row
            print(kind, json.dumps(row), flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], 'w'), indent=1)


if __name__ == '__main__':
    main()
