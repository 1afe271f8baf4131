"""Latent / text-embedding cache of the data feed, on-disk compatible with the reference's
`utils/cache.py:10-133`:

    <dir>/metadata.db   sqlite: fingerprint(value) | items(shard, shard_index) | shard_<N>(offset, size)
    <dir>/shard_<N>.bin concatenated `torch.save` blobs, one per item (item = dict of tensors / python values)

`Cache` keeps the reference's API (`Cache(path, fingerprint, shard_size_gb)`, `len`, `cache[i]`, `add`,
`finalize_current_shard`, `clear`; a changed fingerprint wipes the directory; re-opening appends into a NEW shard) so
`utils/dataset.py:85-161` can use it unchanged.  What is different is the read side, which at MI355X step rates is what
feeds the engine: shards are mmap'd once (no seek + read + copy per item), and `CachePrefetcher` reads ahead on worker
threads into PINNED host memory and issues the host-to-device copies on a side HIP stream, so `train_batch` never waits on
unpickling or on a pageable-memory copy.  (The reference reads with one DataLoader worker and pageable tensors,
utils/dataset.py:1303,1367.)  Measured on the host (tools/cache_feed_rate.py, profiles/r4w_cache_feed_rate_host.json; page cache warm, shuffled order): SDXL-shaped
items 2 100 / s through the reference's reader, this reader and the prefetcher alike (torch.load-bound; the step takes 22.6 / s); Wan-shaped 6.6 MB items 390 / s
(reference: seek + read + BytesIO) vs 730 / s (storages read out of the mapping); the prefetcher's threads cost throughput (330 - 530 / s: the GIL) and buy what they are
for -- a consumer that computes 5 ms per item spends 6 % of that time waiting in next().
"""
import io
import mmap
import os
import pickle
import sqlite3
import struct
import threading
import warnings
import zipfile
from collections import OrderedDict, defaultdict
from pathlib import Path

import torch


# The zero-copy views of a read-only shard mapping make torch.frombuffer warn "The given buffer is not writable" -- read-only is the contract.  Filtered ONCE, here, by
# message: `warnings.catch_warnings()` around every unpickle (round 5) mutated the process-global filter list from the prefetcher's worker threads concurrently.
warnings.filterwarnings('ignore', message='The given buffer is not writable', category=UserWarning)

class _View(io.RawIOBase):
    """Read-only file object over a slice of an mmap (torch.load reads storages straight out of the mapping)."""

    def __init__(self, buf, offset, size):
        self._mv = memoryview(buf)[offset:offset + size]
        self._pos = 0

    def readable(self):
        return True

    def seekable(self):
        return True

    def tell(self):
        return self._pos

    def seek(self, pos, whence=os.SEEK_SET):
        base = {os.SEEK_SET: 0, os.SEEK_CUR: self._pos, os.SEEK_END: len(self._mv)}[whence]
        self._pos = max(0, min(len(self._mv), base + pos))
        return self._pos

    def readinto(self, b):
        n = min(len(b), len(self._mv) - self._pos)
        b[:n] = self._mv[self._pos:self._pos + n]
        self._pos += n
        return n

    def close(self):
        self._mv.release()
        super().close()


class _MappedStorage:
    """persistent-id stand-in of the zero-copy blob reader: where a storage's bytes sit in the shard mapping"""
    __slots__ = ('dtype', 'offset', 'nbytes')

    def __init__(self, dtype, offset, nbytes):
        self.dtype, self.offset, self.nbytes = dtype, offset, nbytes


class _BlobUnpickler(pickle.Unpickler):
    """Unpickles the `data.pkl` of ONE `torch.save` blob (zip container, stored records) whose tensors become VIEWS of the shard mapping: no storage is copied.
    Only what a cache item can hold is admitted (tensors, containers, python scalars / strings): any other global raises, like torch.load(weights_only=True)."""

    def __init__(self, file, mm, records):
        super().__init__(file)
        self._mm, self._records = mm, records

    def find_class(self, mod, name):
        if mod == 'torch._utils' and name == '_rebuild_tensor_v2':
            return self._rebuild
        if mod == 'torch' and name.endswith('Storage') and hasattr(torch, name):
            return getattr(torch, name)
        if mod == 'collections' and name == 'OrderedDict':
            return OrderedDict
        if mod == 'torch' and name == 'Size':
            return torch.Size
        if mod == 'torch' and isinstance(getattr(torch, name, None), torch.dtype):
            return getattr(torch, name)
        raise pickle.UnpicklingError(f'cache blob references {mod}.{name}: not a tensor / container / scalar')

    def persistent_load(self, pid):
        kind, storage_type, key, _location, numel = pid
        assert kind == 'storage', pid
        dtype = torch.uint8 if storage_type is torch.UntypedStorage else storage_type.dtype
        off, size = self._records[key]
        return _MappedStorage(dtype, off, numel * torch.empty((), dtype=dtype).element_size() if storage_type is not torch.UntypedStorage else size)

    def _rebuild(self, storage, storage_offset, size, stride, requires_grad=False, backward_hooks=None, metadata=None):
        if storage.nbytes == 0:
            return torch.empty(tuple(size), dtype=storage.dtype)
        flat = torch.frombuffer(self._mm, dtype=torch.uint8, count=storage.nbytes, offset=storage.offset).view(storage.dtype)
        return torch.as_strided(flat, tuple(size), tuple(stride), storage_offset)


class Cache:
    SMALL_ITEM_BYTES = 1 << 20

    def __init__(self, path, fingerprint, shard_size_gb=1, verbose=False):
        self.path = Path(path)
        self.fingerprint = fingerprint
        self.metadata_db = self.path / 'metadata.db'
        self.shard_size_gb = shard_size_gb
        self.verbose = verbose
        os.makedirs(self.path, exist_ok=True)
        self._lock = threading.Lock()
        self.init()

    def _log(self, msg):
        if self.verbose:
            print(f'[CACHE] {msg}')

    def __len__(self):
        return len(self.items)

    # ------------------------------------------------------------------------------------------------------ reading
    def _shard_map(self, shard_id):
        m = self._maps.get(shard_id)
        if m is None:
            with self._lock:
                m = self._maps.get(shard_id)
                if m is None:
                    f = open(self.path / f'shard_{shard_id}.bin', 'rb')
                    m = (mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ), f)
                    self._maps[shard_id] = m
        return m[0]

    def locate(self, idx):
        """-> (shard id, byte offset, byte size) of item idx."""
        shard_id, shard_index = self.items[idx]
        offset, size = self.shard_metadata[shard_id][shard_index]
        return shard_id, offset, size

    def __getitem__(self, idx):
        assert isinstance(idx, int)
        shard_id, offset, size = self.locate(idx)
        if shard_id == self.shard and self.shard_file is not None:
            self.shard_file.flush()                       # reading back an item of the shard being written
            with open(self.path / f'shard_{shard_id}.bin', 'rb') as f:
                f.seek(offset)
                return torch.load(io.BytesIO(f.read(size)), map_location='cpu')
        m = self._shard_map(shard_id)
        if size <= self.SMALL_ITEM_BYTES:
            # small items (SDXL: 0.27 MB of latents + token ids): torch.load issues dozens of tiny reads per blob, each a Python-level readinto on the view --
            # one C-speed copy of the blob out of the mapping into a BytesIO is cheaper (measured on the host, tools/cache_feed_rate.py)
            return torch.load(io.BytesIO(m[offset:offset + size]), map_location='cpu')
        view = _View(m, offset, size)                     # large items (text-encoder states, video latents): storages are read straight out of the mapping
        try:
            return torch.load(view, map_location='cpu')
        finally:
            view.close()

    def view_item(self, idx):
        """Item idx with every tensor a READ-ONLY VIEW of the shard mapping (zero copies; the views die with the mapping: copy before `close()` / `clear()`).  Reads the
        blob's zip directory and unpickles `data.pkl` with a tensor rebuilder that points into the mapping.  For consumers that copy anyway (CachePrefetcher stages
        into pinned memory): one pass over the bytes instead of torch.load's copy followed by the staging copy -- and that pass runs in ATen, without the GIL.  Falls
        back to `cache[idx]` for an item of the shard still being written."""
        shard_id, offset, size = self.locate(idx)
        if shard_id == self.shard and self.shard_file is not None:
            return self[idx]
        m = self._shard_map(shard_id)
        view = _View(m, offset, size)
        try:
            with zipfile.ZipFile(view) as zf:
                records, pkl = {}, None
                for zi in zf.infolist():
                    if zi.compress_type != zipfile.ZIP_STORED:
                        raise ValueError('compressed record in a torch.save blob')
                    n_name, n_extra = struct.unpack_from('<HH', m, offset + zi.header_offset + 26)
                    data_off = offset + zi.header_offset + 30 + n_name + n_extra
                    name = zi.filename
                    if name.endswith('/data.pkl') or name == 'data.pkl':
                        pkl = (data_off, zi.file_size)
                    elif '/data/' in name:
                        records[name.rsplit('/', 1)[1]] = (data_off, zi.file_size)
            if pkl is None:
                raise ValueError('not a torch.save zip blob')
            return _BlobUnpickler(io.BytesIO(m[pkl[0]:pkl[0] + pkl[1]]), m, records).load()     # ("buffer is not writable": filtered once at import, see below)
        finally:
            view.close()

    # ------------------------------------------------------------------------------------------------- open / reset
    def init(self):
        self.con = sqlite3.connect(self.metadata_db, check_same_thread=False)
        self.con.execute('CREATE TABLE IF NOT EXISTS fingerprint(value)')
        row = self.con.execute('SELECT value FROM fingerprint').fetchone()
        if row is not None:
            self._log(f'Existing cache has fingerprint {row[0]}')
            if self.fingerprint != row[0]:
                self._log('Fingerprint changed, deleting existing cache files')
                self.clear()
                return
        else:
            self.con.execute('INSERT INTO fingerprint VALUES(?)', (self.fingerprint,))
        self.con.execute('CREATE TABLE IF NOT EXISTS items(shard, shard_index)')
        self.items = self.con.execute('SELECT shard, shard_index FROM items').fetchall() or []
        self.shard = max((s for s, _ in self.items), default=-1) + 1        # appends go to a new shard
        self.shard_file = None
        self.shard_metadata = defaultdict(list)
        for (table,) in self.con.execute('SELECT name FROM sqlite_master').fetchall():
            if table.startswith('shard_'):
                self.shard_metadata[int(table.split('_')[-1])] = self.con.execute(f'SELECT offset, size FROM {table}').fetchall()
        self._maps = {}
        self.con.commit()
        self._log(f'Existing cache length: {len(self)}')

    def _close_maps(self):
        for m, f in getattr(self, '_maps', {}).values():
            m.close()
            f.close()
        self._maps = {}

    def clear(self):
        """Delete every cache file, start empty under the current fingerprint."""
        self._close_maps()
        if getattr(self, 'shard_file', None) is not None:
            self.shard_file.close()
            self.shard_file = None
        self.con.close()
        os.remove(self.metadata_db)
        for bin_path in self.path.glob('*.bin'):
            os.remove(bin_path)
        self.init()

    def close(self):
        self.finalize_current_shard()
        self._close_maps()
        self.con.close()

    # ------------------------------------------------------------------------------------------------------ writing
    def create_new_shard(self):
        self.shard_file = open(self.path / f'shard_{self.shard}.bin', 'wb')
        self.shard_table = f'shard_{self.shard}'
        # The shard's table and its rows commit together in finalize_current_shard() (the reference opens the connection with
        # autocommit=False, utils/cache.py:29; Python 3.10's legacy mode would commit a bare CREATE TABLE at once and a process
        # killed mid-shard would leave an empty shard_N table that the next run trips over).
        if not self.con.in_transaction:
            self.con.execute('BEGIN')
        self.con.execute(f'CREATE TABLE {self.shard_table}(offset, size)')
        self.shard_index = 0
        self.offset = 0

    def finalize_current_shard(self):
        if self.shard_file is None:
            return
        self.shard_file.close()
        self.shard_file = None
        self.shard += 1
        self.con.commit()

    def add(self, item):
        if self.shard_file is None:
            self.create_new_shard()
        buffer = io.BytesIO()
        torch.save(item, buffer)
        blob = buffer.getbuffer()
        self.shard_file.write(blob)
        entry = (self.shard, self.shard_index)
        self.items.append(entry)
        self.con.execute('INSERT INTO items VALUES(?, ?)', entry)
        self.shard_index += 1
        size = len(blob)
        self.shard_metadata[self.shard].append((self.offset, size))
        self.con.execute(f'INSERT INTO {self.shard_table} VALUES (?, ?)', (self.offset, size))
        self.offset += size
        if self.shard_file.tell() / 1_000_000_000 >= self.shard_size_gb:
            self.finalize_current_shard()


def _map_tensors(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map_tensors(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map_tensors(v, fn) for v in obj)
    return obj


class CachePrefetcher:
    """Ordered read-ahead over `cache[i] for i in indices`: `workers` threads read up to `depth` items ahead -- tensors as zero-copy views of the shard
    mapping (Cache.view_item), copied ONCE into pinned host memory by ATen (no GIL) -- and, with `device`, enqueue the
    host-to-device copies on a side HIP stream; the consumer's stream waits on the copy event only when it takes the item.
    Iteration order is exactly `indices` (the reference's batch order, utils/dataset.py:347-390, is computed upstream)."""

    def __init__(self, cache, indices, depth=8, workers=None, device=None, pin=None):
        self.cache, self.indices = cache, list(indices)
        if workers is None:
            # one reader thread for small blobs (SDXL: 0.27 MB -- per-item Python work dominates and a second thread only adds GIL hand-offs: 1 870 vs 1 550 items / s),
            # two for large ones (video latents + text states: the copy out of the mapping runs without the GIL and scales)
            probe = [cache.locate(i)[2] for i in self.indices[:8]] if hasattr(cache, 'locate') else []
            workers = 2 if probe and sorted(probe)[len(probe) // 2] >= (1 << 20) else 1
        self.device = torch.device(device) if device is not None else None
        self.pin = (torch.cuda.is_available() if pin is None else pin)
        self.depth, self.workers = max(1, depth), max(1, workers)
        self._copy_stream = torch.cuda.Stream(self.device) if (self.device is not None and self.device.type == 'cuda') else None
        self._slots = {}
        self._cv = threading.Condition()
        self._next_fetch = 0
        self._next_take = 0
        self._stop = False
        self._threads = [threading.Thread(target=self._work, daemon=True) for _ in range(self.workers)]
        for t in self._threads:
            t.start()

    def _stage(self, idx):
        """read item idx and stage it: tensors come as views of the shard mapping (Cache.view_item) and are copied ONCE -- into pinned memory when a GPU will take
        them, into ordinary memory otherwise; the copy runs in ATen without the GIL, so worker threads scale"""
        item = self.cache.view_item(idx) if hasattr(self.cache, 'view_item') else self.cache[idx]
        if self.pin:
            def own(t):
                out = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                out.copy_(t)
                return out
        else:
            def own(t):
                return t.clone(memory_format=torch.contiguous_format)
        item = _map_tensors(item, own)
        event = None
        if self._copy_stream is not None:
            with torch.cuda.stream(self._copy_stream):
                item = _map_tensors(item, lambda t: t.to(self.device, non_blocking=True))
                event = self._copy_stream.record_event()
        return item, event

    def _work(self):
        while True:
            with self._cv:
                while not self._stop and (self._next_fetch >= len(self.indices) or self._next_fetch - self._next_take >= self.depth):
                    if self._next_fetch >= len(self.indices):
                        return
                    self._cv.wait()
                if self._stop:
                    return
                pos = self._next_fetch
                self._next_fetch += 1
            try:
                staged = self._stage(self.indices[pos])
            except BaseException as e:          # surfaced on the consumer thread when it reaches this position
                staged = (e, None)
            with self._cv:
                self._slots[pos] = staged
                self._cv.notify_all()

    def __iter__(self):
        return self

    def __len__(self):
        return len(self.indices)

    def __next__(self):
        with self._cv:
            if self._next_take >= len(self.indices):
                raise StopIteration
            pos = self._next_take
            while pos not in self._slots:
                self._cv.wait()
            item, event = self._slots.pop(pos)
            self._next_take += 1
            self._cv.notify_all()
        if isinstance(item, BaseException):
            raise item
        if event is not None:
            torch.cuda.current_stream(self.device).wait_event(event)
            _map_tensors(item, lambda t: t.record_stream(torch.cuda.current_stream(self.device)) or t)
        return item

    def close(self):
        with self._cv:
            self._stop = True
            self._cv.notify_all()
        for t in self._threads:
            t.join(timeout=5)
