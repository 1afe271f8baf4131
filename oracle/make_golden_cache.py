"""ORACLE fixture generator (test infrastructure): writes a small latent cache with the REFERENCE'S OWN writer
(/root/reference/utils/cache.py, imported unmodified) into tests/golden/cache_ref/, so the product's reader / writer
(diffusion_pipe_amd/cache.py) is pinned against the real on-disk format -- sqlite tables + shard_N.bin blobs.

Run in the build container (the reference tree does not exist on the GPU box):   python oracle/make_golden_cache.py
One shim: the reference calls sqlite3.connect(..., autocommit=False), a Python >= 3.12 keyword; on this image's 3.10 the
keyword is dropped (3.10's default isolation level is the same transactional mode: explicit commit())."""
import importlib.util
import json
import os
import shutil
import sqlite3
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'tests', 'golden', 'cache_ref')


def golden_items():
    """The item sequence both writers are fed (dicts of tensors / python values, like utils/dataset.py:141-155 stores)."""
    g = torch.Generator().manual_seed(1234)
    items = []
    for i in range(7):
        h, w = 8 + 2 * (i % 3), 8 + 2 * (i % 2)
        items.append({'latents': torch.randn(4, h, w, generator=g), 'mask': None if i % 2 else torch.ones(h, w),
                      'caption': f'caption {i}', 'te_embed': torch.randn(5 + i, 16, generator=g).to(torch.bfloat16),
                      'seq_len': torch.tensor(5 + i), 'image_spec': (i, f'/data/img_{i}.png')})
    return items


def main():
    spec = importlib.util.spec_from_file_location('ref_cache', '/root/reference/utils/cache.py')
    ref = importlib.util.module_from_spec(spec)
    real_connect = sqlite3.connect
    if sys.version_info < (3, 12):
        sqlite3.connect = lambda *a, **kw: real_connect(*a, **{k: v for k, v in kw.items() if k != 'autocommit'})
    spec.loader.exec_module(ref)
    shutil.rmtree(OUT, ignore_errors=True)
    cache = ref.Cache(OUT, 'golden-fingerprint-v1', shard_size_gb=9e-6)      # ~9 KB shards -> several items per shard, several shards
    for item in golden_items():
        cache.add(item)
    cache.finalize_current_shard()
    cache.con.close()
    sqlite3.connect = real_connect
    meta = {'torch': torch.__version__, 'items': 7, 'fingerprint': 'golden-fingerprint-v1', 'shard_size_gb': 9e-6,
            'files': {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT))}}
    with open(os.path.join(OUT, 'manifest.json'), 'w') as fh:
        json.dump(meta, fh, indent=1)
    print(meta)


if __name__ == '__main__':
    main()
