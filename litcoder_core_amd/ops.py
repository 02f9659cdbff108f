"""Thin typed wrappers over the C ABI (include/litcoder_hip.h).

torch is used here only as the device-memory container (``torch.empty(..., device=...)``,
``data_ptr()``) and for the current HIP stream; every computation is a call into
liblitcoder_hip.so.  No function here has a CPU path.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib
from ._lib import COL_TILE, K_TILE, LC_F32, LC_F64, LC_MB, LC_NB  # noqa: F401

_checked_devices = set()


def pad_to(n, g):
    return ((int(n) + g - 1) // g) * g


def device(index=None):
    """The torch device the path runs on; raises when no gfx950 GPU is usable."""
    if not torch.cuda.is_available():
        raise _lib.LitcoderHipError("no HIP device visible: litcoder_core_amd runs on MI355X only (no CPU fallback)")
    idx = torch.cuda.current_device() if index is None else int(index)
    if idx not in _checked_devices:
        _lib.call("lc_check_device", idx)
        _checked_devices.add(idx)
    return torch.device("cuda", idx)


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need(t, dtype, what):
    if t.dtype != dtype or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{what}: expected contiguous {dtype} device tensor, got {t.dtype} on {t.device}")


def idx_tensor(rows, length, dev, fill=-1):
    """int32 device index list of ``length`` entries: ``rows`` then ``fill`` padding."""
    h = np.full(length, fill, dtype=np.int32)
    r = np.asarray(rows, dtype=np.int64)
    h[: r.size] = r
    return upload(h, dev)


def filled(shape, dtype, dev, byte=0):
    """torch.empty + hipMemsetAsync on the current stream (0 = zeros, 0xFF = -1 for int32): no framework fill kernel."""
    t = torch.empty(shape, dtype=dtype, device=dev)
    _lib.call("lc_fill_bytes", _p(t), int(byte), t.numel() * t.element_size(), _s())
    return t


def zeros(shape, dtype, dev):
    return filled(shape, dtype, dev, 0)


def idx_matrix(row_sets, length, dev, fill=-1):
    """(len(row_sets), length) int32 device matrix of index lists, one upload."""
    h = np.full((len(row_sets), length), fill, dtype=np.int32)
    for i, rows in enumerate(row_sets):
        r = np.asarray(rows, dtype=np.int64)
        h[i, : r.size] = r
    return upload(h, dev)


import threading

_STAGE = {"buf": None, "off": 0, "lock": threading.Lock(), "events": None}
_STAGE_BYTES = 32 << 20
_STAGE_SEGMENTS = 8                 # the ring is recycled segment by segment


def upload(arr, dev):
    """Small host array -> device through pinned memory, asynchronously on the current stream.  (A pageable
    ``.to(dev)`` / ``torch.tensor(..., device=dev)`` synchronises the stream: with two streams in flight that
    stalls the host behind all the work already queued.)  The pinned bytes come from ONE staging ring of the process,
    bump-allocated: ``tensor.pin_memory()`` takes a block from torch's caching host allocator, which cannot reuse a
    block whose last copy is still queued behind a busy GPU and then calls hipHostMalloc -- measured: an 11 ms host
    stall in the middle of a fit while five 19 KB index lists were uploaded.  The ring is recycled in segments: before a
    segment is written again the copies issued from it on the previous lap are waited for, one event each (round 2
    synchronised the whole DEVICE at every wrap: a hidden stall in a long-lived process)."""
    a = np.ascontiguousarray(arr)
    if dev.type != "cuda":
        return torch.from_numpy(a).to(dev)
    n = a.nbytes
    seg_bytes = _STAGE_BYTES // _STAGE_SEGMENTS
    if n == 0 or n > seg_bytes:
        return torch.from_numpy(a).pin_memory().to(dev, non_blocking=True)
    out = torch.empty(a.shape, dtype=torch.from_numpy(a[:0]).dtype if a.size else torch.float32, device=dev)
    with _STAGE["lock"]:                            # the ring is the process's: fits in several threads share it
        if _STAGE["buf"] is None:
            _STAGE["buf"] = torch.empty(_STAGE_BYTES, dtype=torch.uint8, pin_memory=True)
            _STAGE["events"] = [[] for _ in range(_STAGE_SEGMENTS)]
        off = (_STAGE["off"] + 255) & ~255
        if off >= _STAGE_BYTES:
            off = 0
        seg = off // seg_bytes
        if off + n > (seg + 1) * seg_bytes:         # does not fit the rest of the segment: on to the next one
            seg = (seg + 1) % _STAGE_SEGMENTS
            off = seg * seg_bytes
        if off == seg * seg_bytes:                  # entering the segment: its copies of the previous lap must be done
            for ev in _STAGE["events"][seg]:
                ev.synchronize()
            _STAGE["events"][seg] = []
        _STAGE["off"] = off + n
        stage = _STAGE["buf"][off:off + n]
        stage.numpy()[:] = a.reshape(-1).view(np.uint8)
        out.view(torch.uint8).reshape(-1).copy_(stage, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        _STAGE["events"][seg].append(ev)
    return out


def timing_enable(on=True, only=None):
    """Event timers of the library's kernel classes on / off; ``only``: the slot names to time (the others stay off)."""
    if on and only:
        lib = _lib.load()
        names = [lib.lc_timing_name(s).decode() for s in range(lib.lc_timing_slots())]
        mask = 0
        for n in only:
            mask |= 1 << names.index(n)
        _lib.call("lc_timing_enable_slots", mask)
        return
    _lib.call("lc_timing_enable", int(bool(on)))


def timing_read():
    """{kernel class: (total milliseconds, launches)} since the last read (synchronises)."""
    lib = _lib.load()
    out = {}
    for slot in range(lib.lc_timing_slots()):
        ms, n = ctypes.c_double(0), ctypes.c_int(0)
        _lib.call("lc_timing_read", slot, ctypes.byref(ms), ctypes.byref(n))
        if n.value:
            out[lib.lc_timing_name(slot).decode()] = (ms.value, n.value)
    return out


# ------------------------------------------------------------------ preprocessing
def fir_delay(stim, delays, circpad):
    """stim: (nt, ndim) f32/f64 device tensor -> (nt, ndim*len(delays)) f64 device tensor."""
    dt = LC_F32 if stim.dtype == torch.float32 else LC_F64
    _need(stim, torch.float32 if dt == LC_F32 else torch.float64, "fir_delay")
    nt, ndim = stim.shape
    nd = len(delays)
    out = torch.empty((nt, ndim * nd), dtype=torch.float64, device=stim.device)
    arr = (ctypes.c_int64 * max(nd, 1))(*[int(d) for d in delays])
    _lib.call("lc_fir_delay", _p(stim), dt, nt, ndim, ndim, arr, nd, int(bool(circpad)), _p(out), ndim * nd, _s())
    return out


def lanczos_interp(data, oldtime, newtime, cutoff, window, rectify):
    dt = LC_F32 if data.dtype == torch.float32 else LC_F64
    _need(data, torch.float32 if dt == LC_F32 else torch.float64, "lanczos_interp")
    _need(oldtime, torch.float64, "lanczos_interp oldtime")
    _need(newtime, torch.float64, "lanczos_interp newtime")
    n_old, D = data.shape
    n_new = newtime.numel()
    ld_out = 2 * D if rectify else D
    out = torch.empty((n_new, ld_out), dtype=torch.float64, device=data.device)
    _lib.call("lc_lanczos_interp", _p(data), dt, n_old, D, D, _p(oldtime), _p(newtime), n_new, float(cutoff),
              float(window), int(bool(rectify)), _p(out), ld_out, _s())
    return out


def story_table(fields, dev):
    """A table of per-story records for the batched preprocessing kernels: ``fields`` = [(numpy dtype, values), ...] in
    the C struct's field order (int64 / float64 / int32 members; an int32 tail is padded to the 8-byte record stride)."""
    n = len(fields[0][1])
    names = [f"f{i}" for i in range(len(fields))]
    dt = np.dtype({"names": names, "formats": [f[0] for f in fields]}, align=True)
    rec = np.zeros(n, dtype=dt)
    for name, (_, vals) in zip(names, fields):
        rec[name] = vals
    return upload(rec.view(np.uint8).reshape(n, dt.itemsize), dev), dt.itemsize


def lanczos_interp_stories(data, oldtimes, newtimes, window, cutoff_mult, rectify):
    """lanczosinterp2D for many stories in ONE launch (lc_lanczos_interp_stories).  ``data``: (sum n_old, D) f32|f64 device
    matrix, the stories' samples concatenated; ``oldtimes`` / ``newtimes``: per story host float64 arrays.  Returns the
    (sum n_new, D or 2 D) f64 device matrix of the stories' outputs, concatenated, and the output row offsets."""
    dt = LC_F32 if data.dtype == torch.float32 else LC_F64
    _need(data, torch.float32 if dt == LC_F32 else torch.float64, "lanczos_interp_stories")
    dev = data.device
    old = [np.asarray(t, dtype=np.float64).reshape(-1) for t in oldtimes]
    new = [np.asarray(t, dtype=np.float64).reshape(-1) for t in newtimes]
    n_old = np.asarray([len(t) for t in old], dtype=np.int64)
    n_new = np.asarray([len(t) for t in new], dtype=np.int64)
    old_off = np.concatenate([[0], np.cumsum(n_old)])
    new_off = np.concatenate([[0], np.cumsum(n_new)])
    if int(old_off[-1]) != data.shape[0]:
        raise RuntimeError(f"shape mismatch: {data.shape[0]} sample rows, {int(old_off[-1])} sample times")
    with np.errstate(all="ignore"):                     # (a story with < 2 output times: nan cutoff, like the reference)
        cutoff = np.asarray([1.0 / np.mean(np.diff(t)) * cutoff_mult if len(t) else 0.0 for t in new], dtype=np.float64)
    # the bisected window needs (tn - ot[j]) * cutoff to be non-increasing in j: sorted sample times AND a positive finite
    # cutoff -- decreasing output times give a negative one, fewer than two a NaN (lanczosfun then yields NaN weights, which
    # only the full scan reproduces: ADVICE r4)
    ordered = np.asarray([int(bool(np.all(np.diff(t) >= 0)) and bool(np.isfinite(c) and c > 0)) for t, c in zip(old, cutoff)],
                         dtype=np.int32)
    table, stride = story_table([(np.int64, old_off[:-1]), (np.int64, n_old), (np.int64, new_off[:-1]), (np.float64, cutoff),
                                 (np.int32, ordered)], dev)
    assert stride == 40
    row_story = upload(np.repeat(np.arange(len(new), dtype=np.int32), n_new), dev)
    d_old = upload(np.concatenate(old) if old else np.zeros(0), dev)
    d_new = upload(np.concatenate(new) if new else np.zeros(0), dev)
    D = data.shape[1]
    ld_out = 2 * D if rectify else D
    out = torch.empty((int(new_off[-1]), ld_out), dtype=torch.float64, device=dev)
    _lib.call("lc_lanczos_interp_stories", _p(data), dt, D, data.stride(0), _p(d_old), _p(d_new), int(new_off[-1]),
              _p(row_story), _p(table), len(new), float(window), int(bool(rectify)), _p(out), ld_out, _s())
    return out, new_off


def story_design(feat, in_off, n_in, a, b, out_row0, delays, X):
    """The float32 design matrix of a story-structured fit in one launch (lc_story_design_f32): FIR delays + trim +
    per-story zs + nan_to_num + cast.  ``feat``: (sum n_in, ndim) f64 device matrix (stories concatenated by rows);
    per story its row offset / count, the trimmed range [a, b) of its delayed rows and its first row in ``X``."""
    _need(feat, torch.float64, "story_design")
    table, stride = story_table([(np.int64, in_off), (np.int64, n_in), (np.int64, a), (np.int64, b), (np.int64, out_row0)],
                                feat.device)
    assert stride == 40
    nd = len(delays)
    arr = (ctypes.c_int64 * max(nd, 1))(*[int(d) for d in delays])
    _lib.call("lc_story_design_f32", _p(feat), feat.shape[1], feat.stride(0), _p(table), len(in_off), arr, nd, _p(X),
              X.stride(0), _s())
    return X


def host_zscore_story(block):
    """float32 (rows, cols) = fl32(zs(block)) on the HOST, exactly as a z-scored upload job stages a story
    (lc_host_zscore_story; no device involved -- the CPU tests compare it with numpy bit for bit)."""
    (_, b), = HostRows([block]).blocks
    out = np.empty(b.shape, dtype=np.float32)
    item = b.dtype.itemsize
    _lib.call("lc_host_zscore_story", ctypes.c_void_p(b.ctypes.data), LC_F64 if b.dtype == np.float64 else LC_F32,
              b.strides[0] // item if b.shape[0] > 1 else max(b.shape[1], 1), b.shape[0], b.shape[1],
              ctypes.c_void_p(out.ctypes.data), max(b.shape[1], 1))
    return out


def sinc_interp(data, oldtime, newtime, cutoff, window, causal, renorm):
    dt = LC_F32 if data.dtype == torch.float32 else LC_F64
    _need(data, torch.float32 if dt == LC_F32 else torch.float64, "sinc_interp")
    n_old, D = data.shape
    n_new = newtime.numel()
    out = torch.empty((n_new, D), dtype=torch.float64, device=data.device)
    _lib.call("lc_sinc_interp", _p(data), dt, n_old, D, D, _p(oldtime), _p(newtime), n_new, float(cutoff),
              float(window), int(bool(causal)), int(bool(renorm)), _p(out), D, _s())
    return out


def segment_reduce(data, seg, idx, how):
    """data (n, D) f32|f64; seg (n_seg+1) int64 offsets into idx (int32 row numbers); how: 0 mean, 1 sum, 2 last."""
    dt = LC_F32 if data.dtype == torch.float32 else LC_F64
    _need(data, torch.float32 if dt == LC_F32 else torch.float64, "segment_reduce")
    n_seg = seg.numel() - 1
    D = data.shape[1]
    out = torch.empty((n_seg, D), dtype=torch.float64, device=data.device)
    _lib.call("lc_segment_reduce", _p(data), dt, D, D, _p(seg), _p(idx), n_seg, int(how), _p(out), D, _s())
    return out


# ------------------------------------------------------------------ casts / gathers
_UPLOAD = {"pinned": None, "pool": None, "stream": {}}
_UPLOAD_CHUNK = 16 << 20            # bytes per pinned staging chunk
_UPLOAD_DEPTH = 12                  # chunks in flight
_UPLOAD_LOCK = threading.Lock()     # the ring is the process's: one upload job at a time owns it


_UPLOAD_THREADS_ZS = 24             # staging threads of such an upload (capped at half the cores; measured: 24 on the
                                    # data's NUMA node beat 48 and 96, profiles/r04_upload_probe*.txt)
_UPLOAD_DEPTH_ZS = 32               # ... of an upload that z-scores stories on the way (three passes per chunk on the host:
                                    # more threads in flight to keep the link busy)


def _upload_ring(depth=None):
    """Pinned staging chunks of the process (created on first use; grown when a deeper ring is asked for)."""
    depth = _UPLOAD_DEPTH if depth is None else int(depth)
    if _UPLOAD["pinned"] is None:
        _UPLOAD["pinned"] = []
    while len(_UPLOAD["pinned"]) < depth:
        _UPLOAD["pinned"].append(torch.empty(_UPLOAD_CHUNK, dtype=torch.uint8, pin_memory=True))
    return _UPLOAD["pinned"][:depth], None


def misc_pool():
    """One worker thread for slow host-side chores that must not queue behind the staging threads (hipHostMalloc of a
    result buffer)."""
    if _UPLOAD.get("misc") is None:
        from concurrent.futures import ThreadPoolExecutor
        _UPLOAD["misc"] = ThreadPoolExecutor(max_workers=1)
    return _UPLOAD["misc"]


def upload_stream(dev):
    """The stream host-to-device panel copies run on (one per device, for the life of the process)."""
    key = (dev.type, dev.index)
    if key not in _UPLOAD["stream"]:
        _UPLOAD["stream"][key] = torch.cuda.Stream(device=dev)
    return _UPLOAD["stream"][key]


class HostRows:
    """Row blocks of host matrices with the same column count, seen as one (rows, cols) matrix without concatenating
    them (train/test mode hands over the training and the test targets as two arrays; a story-structured fit one block
    per story).  ``zscore``: every block is ONE story and is z-scored on its way to the device (utils.zs as the
    trainer applies it per story, trainer.py:235-257 -- LC_UPLOAD_ZSCORE jobs of the native uploader)."""

    def __init__(self, blocks, zscore=False):
        self.blocks = []
        self.zscore = bool(zscore)
        r = 0
        for b in blocks:
            b = np.asarray(b)
            if b.ndim != 2:
                raise ValueError("expected 2-D arrays")
            if b.dtype not in (np.float32, np.float64):
                b = b.astype(np.float64)
            item = b.dtype.itemsize
            if b.shape[1] and (b.strides[1] != item or (b.shape[0] > 1 and (
                    b.strides[0] < b.shape[1] * item or b.strides[0] % item))):
                # anything the staging threads cannot walk as rows of a positive whole-element stride (a transposed,
                # row-reversed or broadcast view: the reference's torch.tensor(x) accepts them all) is copied once
                b = np.ascontiguousarray(b)
            self.blocks.append((r, b))
            r += b.shape[0]
        cols = {b.shape[1] for _, b in self.blocks}
        if len(cols) != 1:
            raise RuntimeError(f"shape mismatch: row blocks with {sorted(cols)} columns")
        self.shape = (r, cols.pop())


class _UploadJob(ctypes.Structure):
    """lc_upload_job of include/litcoder_hip.h."""
    _fields_ = [("src", ctypes.c_void_p), ("ld_src", ctypes.c_int64), ("dtype", ctypes.c_int), ("rows", ctypes.c_int64),
                ("c0", ctypes.c_int64), ("c1", ctypes.c_int64), ("dst", ctypes.c_void_p), ("ld_dst", ctypes.c_int64),
                ("dst_row0", ctypes.c_int64), ("transform", ctypes.c_int)]


LC_UPLOAD_CAST, LC_UPLOAD_ZSCORE = 0, 1


class PanelUploader:
    """Host matrices -> zero-padded f32 device matrices, in column panels, on native background threads
    (lc_upload_start, csrc/lc_upload.hip).

    The caller's arrays are pageable: a plain copy stages them through the driver at ~10 GB/s.  Here a panel (a column
    range of one host matrix) is cut into row chunks; native staging threads cast each chunk to float32 straight into
    page-locked memory (the cast of ``torch.tensor(x, dtype=torch.float32)``, nested_cv.py:99-100, done on the host so
    that 4 bytes per value cross PCIe) and issue its 2-D copy into the panel's columns of the destination on the upload
    stream.  ``jobs``: [(host matrix | HostRows, destination (rows, ld) f32 device matrix, c0, c1)] in upload order --
    e.g. the design matrix, then the panels of the targets.  ``wait(j, stream)`` blocks the HOST until every copy of job j
    has been issued and makes ``stream`` wait for them on the device.  (Python threads staged the chunks until round 3:
    beside a main thread that queues a thousand launches the interpreter lock made both crawl.)"""

    def __init__(self, jobs, dev, after=None):
        self.jobs = [(h if isinstance(h, HostRows) else HostRows([h]), d, int(c0), int(c1)) for h, d, c0, c1 in jobs]
        self.dev = dev
        self.stream = upload_stream(dev)
        if after is not None:
            self.stream.wait_event(after)               # e.g. the zero fill of the destination's padding
        native, self._natives = [], []
        for host, dst, c0, c1 in self.jobs:
            if dst.dtype != torch.float32 or not dst.is_cuda or dst.stride(1) != 1:
                raise ValueError("upload destination must be a row-major f32 device matrix")
            if not (0 <= c0 < c1 <= host.shape[1] and c1 <= dst.shape[1] and host.shape[0] <= dst.shape[0]):
                raise ValueError("upload panel outside the source / destination matrix")
            first = len(native)
            for row0, blk in host.blocks:
                item = blk.dtype.itemsize
                if host.zscore and blk.shape[0] * 4 > _UPLOAD_CHUNK:
                    raise ValueError("a z-scored story block has more rows than a staging slot holds values")
                native.append(_UploadJob(blk.ctypes.data, blk.strides[0] // item if blk.shape[0] > 1 else max(blk.shape[1], 1),
                                         LC_F64 if blk.dtype == np.float64 else LC_F32, blk.shape[0], c0, c1,
                                         dst.data_ptr(), dst.stride(0), row0,
                                         LC_UPLOAD_ZSCORE if host.zscore else LC_UPLOAD_CAST))
            self._natives.append(list(range(first, len(native))))     # the panel = its row blocks' native jobs
        self._native = (_UploadJob * len(native))(*native)
        pinned, _ = _upload_ring(_UPLOAD_DEPTH_ZS if any(h.zscore for h, _, _, _ in self.jobs) else None)
        self._slots = (ctypes.c_void_p * len(pinned))(*[p.data_ptr() for p in pinned])
        # device staging slots, one per host slot (created once per device): chunks cross the link contiguously
        key = (dev.type, dev.index)
        dslots = _UPLOAD.setdefault("dev_slots", {}).setdefault(key, [])
        use_dev = os.environ.get("LITCODER_AMD_UPLOAD_DEVICE_STAGING", "1") != "0"
        while use_dev and len(dslots) < len(pinned):
            dslots.append(torch.empty(_UPLOAD_CHUNK, dtype=torch.uint8, device=dev))
        self._dslots = (ctypes.c_void_p * len(pinned))(*[d.data_ptr() for d in dslots[:len(pinned)]]) if use_dev else None
        self._handle = ctypes.c_void_p()
        self._done = False
        # staging threads: half the cores, shared out over the processes of the node (one per GPU under torchrun)
        per_node = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1") or 1))
        zs = any(h.zscore for h, _, _, _ in self.jobs)   # (z-scored chunks are shared by several threads: more of them)
        n_threads = max(2, min(_UPLOAD_THREADS_ZS if zs else len(pinned), (os.cpu_count() or 4) // 2 // per_node))
        if os.environ.get("LITCODER_AMD_UPLOAD_THREADS"):      # (tuning probes: tools/upload_probe.py)
            n_threads = max(1, int(os.environ["LITCODER_AMD_UPLOAD_THREADS"]))
        _UPLOAD_LOCK.acquire()                          # the staging ring is the process's: one upload at a time owns it
        try:
            _lib.call("lc_upload_start_staged", ctypes.cast(self._native, ctypes.c_void_p), len(native),
                      ctypes.cast(self._slots, ctypes.c_void_p),
                      ctypes.cast(self._dslots, ctypes.c_void_p) if self._dslots is not None else None, len(pinned),
                      _UPLOAD_CHUNK, n_threads,
                      dev.index if dev.index is not None else torch.cuda.current_device(),
                      ctypes.c_void_p(self.stream.cuda_stream), ctypes.byref(self._handle))
        except BaseException:
            _UPLOAD_LOCK.release()
            raise
        self.error = None
        # NOT a daemon thread: a process that ends while an upload is still in flight (an exception on the caller's way out,
        # a script's last statement) must wait for it -- the interpreter joins this thread before it finalises.  As a daemon
        # it came back from lc_upload_finish into a finalising interpreter, which ends such a thread with pthread_exit: a
        # forced unwind through foreign frames, "terminate called without an active exception", SIGABRT at exit (seen once
        # in some hundred fuzz-worker processes, round 6)
        self.thread = threading.Thread(target=self._finish, name="lc-upload-finish", daemon=False)
        self.thread.start()

    def _finish(self):
        """Joins the native threads (blocked in C, interpreter lock released) and hands the staging ring back."""
        try:
            _lib.call("lc_upload_finish", self._handle)
        except BaseException as exc:  # noqa: BLE001 -- handed to whoever joins
            self.error = exc
        finally:
            self._done = True
            _UPLOAD_LOCK.release()

    def wait(self, j, stream=None):
        """Host: until job j's copies are all issued; device: ``stream`` (default: current) waits for them."""
        stream = stream or torch.cuda.current_stream()
        if self._done and self.error is not None:
            raise self.error
        for k in self._natives[j]:
            _lib.call("lc_upload_wait", self._handle, k, ctypes.c_void_p(stream.cuda_stream))

    def join(self):
        self.thread.join()
        if self.error is not None:
            raise self.error

    def __del__(self):
        try:
            if self._handle:
                self.thread.join()
                _lib.load().lc_upload_free(self._handle)
                self._handle = ctypes.c_void_p()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


class TargetsInFlight:
    """Host targets (HostRows) on their way into their (T, pad(V)) float32 device matrix, column panel by column panel,
    started BEFORE the engine that will fit them exists -- harness.StoryPipeline puts the brain data on the link first
    and builds the design (Lanczos resampling, FIR delays, z-scoring: ~8 ms of host and device work) beside it, where the
    engine's own uploader starts only when the design is there.  ``lead``: [(host rows, destination, c0, c1)] jobs that
    cross the link FIRST through the same staging ring (the word-level features the design is built from);
    ``wait_lead`` makes a stream wait for them.  RidgeCVEngine takes the object in place of the host targets."""

    def __init__(self, host_rows, dev, panels, lead=()):
        self.host = host_rows if isinstance(host_rows, HostRows) else HostRows([host_rows])
        self.shape = self.host.shape
        T, V = self.shape
        if T < 1 or V < 1:
            raise ValueError("targets in flight need at least one row and one column")
        self.Vp = pad_to(V, COL_TILE)
        self.panels = [(int(a), int(b)) for a, b in panels]
        self.buffer = torch.empty((T, self.Vp), dtype=torch.float32, device=dev)
        zero_cols(self.buffer, V, self.Vp)
        zeroed = torch.cuda.Event()
        zeroed.record()
        self.n_lead = len(lead)
        self.uploader = PanelUploader(list(lead) + [(self.host, self.buffer, a, b) for a, b in self.panels], dev, after=zeroed)

    def wait_lead(self, stream=None):
        for j in range(self.n_lead):
            self.uploader.wait(j, stream)

    def __del__(self):
        # dropped before a fit took it to its end (an exception on the caller's side): the staging threads and the copies
        # they queued still write into the buffer -- it goes back to the allocator only when they are done
        try:
            up = getattr(self, "uploader", None)
            if up is not None:
                up.thread.join()
                up.stream.synchronize()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


def upload_f32(host, ld, dev, rows_pad=None):
    """Host (rows, cols) real array -> zero-padded (rows_pad or rows, ld) f32 device buffer, ordered on the current
    stream.  float64 input is cast to float32 on the host while it is staged (PanelUploader), mirroring
    ``torch.tensor(x, dtype=torch.float32)`` (nested_cv.py:99-100)."""
    host = host if isinstance(host, HostRows) else HostRows([host])
    rows, cols = host.shape
    out = zeros((rows_pad or rows, ld), torch.float32, dev)
    if rows == 0 or cols == 0:
        return out
    if cols * 4 > _UPLOAD_CHUNK:                        # absurdly wide rows: the plain path
        for row0, blk in host.blocks:
            out[row0:row0 + blk.shape[0], :cols].copy_(torch.from_numpy(np.ascontiguousarray(blk, dtype=np.float32)))
        return out
    zeroed = torch.cuda.Event()
    zeroed.record()
    up = PanelUploader([(host, out, 0, cols)], dev, after=zeroed)
    up.wait(0)
    up.join()
    return out


def download_cols(src, host, c0, V, stream):
    """Columns [0, V) of the device matrix view ``src`` (rows, >= V) -> columns [c0, c0 + V) of the page-locked host
    matrix ``host``, as one 2-D copy on ``stream``."""
    _lib.call("lc_memcpy2d_async", ctypes.c_void_p(host.data_ptr() + c0 * 4), host.stride(0) * 4, _p(src), src.stride(0) * 4,
              V * 4, src.shape[0], 1, ctypes.c_void_p(stream.cuda_stream))


def zero_cols(t, c0, c1):
    """Zero columns [c0, c1) of a 2-D f32 device matrix on the current stream (hipMemset2DAsync, no framework kernel)."""
    if c1 > c0:
        _lib.call("lc_fill2d_bytes", ctypes.c_void_p(t.data_ptr() + c0 * 4), t.stride(0) * 4, 0, (c1 - c0) * 4, t.shape[0], _s())


def cast_f64_f32(src, dst, rows, cols):
    """(rows, cols) float64 device view -> float32 device view (row strides taken from the tensors)."""
    _lib.call("lc_cast_f64_f32", _p(src), src.stride(0), _p(dst), dst.stride(0), rows, cols, _s())


def gather(src, ld_src, rows, n_rows, cols, n_cols, out, live=None):
    """``live``: device int32 -- only the first ``*live`` columns (whole 256-column tiles) are touched (the refinement's panel)."""
    _lib.call("lc_gather_f32", _p(src), ld_src, _p(rows), n_rows, _p(cols), n_cols, _p(out), out.stride(0), _p(live), _s())
    return out


def scatter_axpy(w, n_rows, cols, n_cols, scale, acc):
    _lib.call("lc_scatter_axpy_f32", _p(w), w.stride(0), n_rows, _p(cols), n_cols, float(scale), _p(acc),
              acc.stride(0), _s())


def invert_perm(cols, n_cols, base, pos):
    """pos[cols[j]] = base + j for the live entries of the alpha-sorted column list (lc_invert_perm)."""
    _lib.call("lc_invert_perm", _p(cols), n_cols, int(base), _p(pos), _s())


def combine_folds(parts, n_rows, n_cols, out):
    """out[r, v] = sum_f scale_f * w_f[r, pos_f[v]] over ``parts`` = [(w_f (rows, ld_f) f32, pos_f (>= n_cols,) int32,
    scale_f)], folds in order (lc_combine_folds_f32); out: (rows, >= n_cols) f32 device view."""
    n = len(parts)
    w = (ctypes.c_void_p * n)(*[p_[0].data_ptr() for p_ in parts])
    ld = (ctypes.c_int64 * n)(*[int(p_[0].stride(0)) for p_ in parts])
    pos = (ctypes.c_void_p * n)(*[p_[1].data_ptr() for p_ in parts])
    sc = (ctypes.c_float * n)(*[float(p_[2]) for p_ in parts])
    _lib.call("lc_combine_folds_f32", w, ld, pos, sc, n, n_rows, n_cols, _p(out), out.stride(0), _s())
    return out


# ------------------------------------------------------------------ column statistics
def col_mean_std(x, rows, n_rows, n_cols):
    mean = torch.empty(n_cols, dtype=torch.float32, device=x.device)
    std = torch.empty(n_cols, dtype=torch.float32, device=x.device)
    _lib.call("lc_col_mean_std_f32", _p(x), x.stride(0), _p(rows), n_rows, n_cols, _p(mean), _p(std), _s())
    return mean, std


def col_normalize_(x, n_rows, n_cols, mean, std, eps=1e-8):
    _lib.call("lc_col_normalize_f32", _p(x), x.stride(0), n_rows, n_cols, _p(mean), _p(std), float(eps), _s())


def zscore_story(x, rows, cols, nan_to_num, out):
    """x: f64 device view (rows, >=cols); out: f64 device view, same shape; utils.zs semantics."""
    _lib.call("lc_zscore_story_f64", _p(x), x.stride(0), rows, cols, int(bool(nan_to_num)), _p(out), out.stride(0), _s())
    return out


def pearson_cols(a, b, n, V):
    r = torch.empty(V, dtype=torch.float64, device=a.device)
    _lib.call("lc_pearson_cols", _p(a), a.stride(0), _p(b), b.stride(0), n, V, _p(r), _s())
    return r


def pearson_cols_gather(y, rows, cols, b, n, V):
    """Pearson r of y[rows][:, cols] against b (n, V) per column, y read in place (lc_pearson_cols_gather)."""
    r = torch.empty(V, dtype=torch.float64, device=y.device)
    _lib.call("lc_pearson_cols_gather", _p(y), y.stride(0), _p(rows), _p(cols), _p(b), b.stride(0), n, V, _p(r), _s())
    return r


def pearson_pvalues(r, V, n):
    p = torch.empty(V, dtype=torch.float64, device=r.device)
    _lib.call("lc_pearson_pvalues", _p(r), V, n, _p(p), _s())
    return p


# ------------------------------------------------------------------ small dense fp64
GRAM_MFMA_MIN_T = 2560             # from here on the Gram matrix goes through the fp64 MFMA (lc_gram_f64_mfma: 0.87 vs 1.17 ms at T 3000, no gain at 2226)


def gram(x, T, p):
    k = torch.empty((T, T), dtype=torch.float64, device=x.device)
    if T >= GRAM_MFMA_MIN_T:
        work = torch.empty(T * pad_to(p, 16), dtype=torch.float64, device=x.device)
        _lib.call("lc_gram_f64_mfma", _p(x), x.stride(0), T, p, _p(work), _p(k), T, _s())
    else:
        _lib.call("lc_gram_f64", _p(x), x.stride(0), T, p, _p(k), T, _s())
    return k


def lambda_max(k, rows, F, N, steps):
    work = torch.empty(F * (3 * N + 2 * steps + 8), dtype=torch.float64, device=k.device)
    out = torch.empty(F, dtype=torch.float64, device=k.device)
    _lib.call("lc_lambda_max", _p(k), k.stride(0), 0, _p(rows), F, N, steps, _p(work), _p(out), _s())
    return out


def lambda_max_masked(k, T, member, F, steps, out=None, use_mfma=True, tol=0.0):
    """lambda_max of K[I_f, I_f] for F <= 32 row sets given as bit f of member[i] (int32 tensor of T words).  ``tol`` > 0:
    a system stops once its top Ritz value has moved by <= tol (relative) over 8 steps (lc_lambda_max_masked)."""
    work = torch.empty(F * (3 * T + 2 * steps + 8) + 16 * 32 * T, dtype=torch.float64, device=k.device)
    if out is None:
        out = torch.empty(F, dtype=torch.float64, device=k.device)
    _lib.call("lc_lambda_max_masked", _p(k), k.stride(0), T, _p(member), F, steps, float(tol), _p(work), _p(out),
              int(bool(use_mfma)), _s())
    return out


def lambda_max_dense(k, ldk, k_stride, F, N, n, steps, tol=0.0):
    """lambda_max of the leading n x n blocks of F matrices (matrix f at k + f k_stride doubles, row stride ldk): the
    streaming matvec of lc_lambda_max_dense.  N: padded vector length."""
    work = torch.empty(F * (3 * N + 2 * steps + 8), dtype=torch.float64, device=k.device)
    out = torch.empty(F, dtype=torch.float64, device=k.device)
    _lib.call("lc_lambda_max_dense", _p(k), ldk, k_stride, F, N, n, steps, float(tol), _p(work), _p(out), _s())
    return out


def penalties(lmax, F, alphas, normalpha):
    A = alphas.numel()
    a2 = torch.empty(F * A, dtype=torch.float64, device=alphas.device)
    _lib.call("lc_penalties", _p(lmax), F, _p(alphas), A, int(bool(normalpha)), _p(a2), _s())
    return a2


def batch_assemble(k, tr, va, rhs, a2, F, A, N, M, aug):
    _lib.call("lc_batch_assemble", _p(k), k.stride(0), _p(tr), _p(va), _p(rhs), _p(a2), F, A, N, M, _p(aug), _s())


def batch_assemble_sel(k, tr, va, rhs, a2, sys, B, A, N, M, aug, k_fold_stride=0, ldk=None):
    """System j of the batch = grid system sys[j] = fold * A + alpha (see lc_batch_assemble_sel); ``k_fold_stride``
    (elements) > 0: fold f's top block comes from the matrix at k + f * stride (primal form)."""
    _lib.call("lc_batch_assemble_sel", _p(k), k.stride(-2) if ldk is None else ldk, int(k_fold_stride), _p(tr), _p(va),
              _p(rhs), _p(a2), _p(sys), B, A, N, M, _p(aug), _s())


def gather_transpose_f32(x, rows, F, N, p, p_pad):
    """(F * p_pad, N) f32: block f = X[rows[f]]' zero-padded (rows: (F, N) int32 device, -1 = zero column)."""
    out = torch.empty((F * p_pad, N), dtype=torch.float32, device=x.device)
    _lib.call("lc_gather_transpose_f32", _p(x), x.stride(0), _p(rows), F, N, p, p_pad, _p(out), _s())
    return out


def gram_blocks(xt, n_blocks, rows_per, depth):
    """(n_blocks, rows_per, rows_per) f64: block b = Xt_b Xt_b' for the row blocks of xt (f32, ld = xt.stride(0))."""
    g = torch.empty((n_blocks, rows_per, rows_per), dtype=torch.float64, device=xt.device)
    _lib.call("lc_gram_blocks_f64", _p(xt), xt.stride(0), n_blocks, rows_per, depth, _p(g), _s())
    return g


def gather_rows_f64(x, rows, F, M, p, N):
    """(F, M, N) f64 rows of X by index list (-1 zero row, -(2 + c) unit row e_c)."""
    out = torch.empty((F, M, N), dtype=torch.float64, device=x.device)
    _lib.call("lc_gather_rows_f64", _p(x), x.stride(0), _p(rows), F, M, p, N, _p(out), _s())
    return out


# ---- primal form for a handful of features (csrc/lc_primal.hip)
PRIMAL_CHUNK = 2048                # rows per partial block product


def primal_pad(p):
    return int(_lib.load().lc_primal_pad(int(p)))


def xty(x, p, y, V, rows, nrows, shrow, n_sets, part=None):
    """(n_sets, RS, p_pad + 2, V) f64 partial block products X'(Y - shift) of the row sets (see lc_xty_f64)."""
    ldr = rows.shape[-1]
    RS = -(-ldr // PRIMAL_CHUNK)
    if part is None:
        part = torch.empty((n_sets, RS, primal_pad(p) + 2, V), dtype=torch.float64, device=y.device)
    _lib.call("lc_xty_f64", _p(x), x.stride(0), p, _p(y), y.stride(0), V, _p(rows), ldr, _p(nrows), _p(shrow), n_sets,
              PRIMAL_CHUNK, RS, _p(part), _s())
    return part


def primal_set_stats(x, p, rows, nrows, n_sets):
    PT = primal_pad(p)
    out = torch.empty((n_sets, PT + 2 * PT * PT), dtype=torch.float64, device=x.device)
    _lib.call("lc_primal_set_stats", _p(x), x.stride(0), p, _p(rows), rows.shape[-1], _p(nrows), n_sets, _p(out), _s())
    return out


def primal_gsys(xstat, sysdef, n_sys, p):
    PT = primal_pad(p)
    out = torch.empty((n_sys, PT, PT), dtype=torch.float64, device=xstat.device)
    _lib.call("lc_primal_gsys", _p(xstat), _p(sysdef), n_sys, p, _p(out), _s())
    return out


def primal_inverse(gsys, a2, n_sys, A, p):
    PT = primal_pad(p)
    pinv = torch.empty((n_sys * A, PT, PT), dtype=torch.float64, device=gsys.device)
    info = torch.empty(n_sys * A, dtype=torch.int32, device=gsys.device)
    _lib.call("lc_primal_inverse", _p(gsys), _p(a2), n_sys, A, p, _p(pinv), _p(info), _s())
    return pinv, info


def primal_scores(part, nrows, shrow, y, V, src, xstat, pinv, F, A, p, scores):
    _lib.call("lc_primal_scores", _p(part), part.shape[1], PRIMAL_CHUNK, _p(nrows), _p(shrow), _p(y), y.stride(0), V,
              _p(src), _p(xstat), _p(pinv), F, A, p, _p(scores), scores.stride(0), _s())
    return scores


def primal_refit(part, nrows, shrow, y, V, set_train, set_test, xstat, pinv, best, p, scale, W, r):
    _lib.call("lc_primal_refit", _p(part), part.shape[1], PRIMAL_CHUNK, _p(nrows), _p(shrow), _p(y), y.stride(0), V,
              set_train, set_test, _p(xstat), _p(pinv), _p(best), p, float(scale), _p(W), W.stride(0), _p(r), _s())
    return r


def lambda_max_strided(k, ldk, k_stride, rows, F, N, steps):
    work = torch.empty(F * (3 * N + 2 * steps + 8), dtype=torch.float64, device=k.device)
    out = torch.empty(F, dtype=torch.float64, device=k.device)
    _lib.call("lc_lambda_max", _p(k), ldk, k_stride, _p(rows), F, N, steps, _p(work), _p(out), _s())
    return out


def masked_stream(mask_words):
    """torch ExternalStream over a HIP stream limited to the CUs set in mask_words (sequence of uint32)."""
    import ctypes as ct
    words = (ct.c_uint32 * len(mask_words))(*[int(w) & 0xFFFFFFFF for w in mask_words])
    out = ct.c_void_p()
    _lib.call("lc_stream_create_cu_mask", ct.cast(words, ct.c_void_p), len(mask_words), ct.cast(ct.byref(out), ct.c_void_p))
    return torch.cuda.ExternalStream(out.value, device=device())


class CholOptions(ctypes.Structure):
    """lc_chol_options of include/litcoder_hip.h: per-call variants of the batched Cholesky (defaults: 512, 2, 1, 0)."""
    _fields_ = [("outer_block", ctypes.c_int), ("big_kernel", ctypes.c_int), ("fused_steps", ctypes.c_int),
                ("left_deep", ctypes.c_int), ("persistent", ctypes.c_int)]


def chol_options(outer_block=512, big_kernel=2, fused_steps=True, left_deep=False, persistent=1):
    return CholOptions(int(outer_block), int(big_kernel), int(bool(fused_steps)), int(bool(left_deep)), int(persistent))


def batch_chol_solve(aug, B, N, M, h, slot=None, options=None):
    linv = torch.empty((B, N // LC_NB, LC_NB, LC_NB), dtype=torch.float64, device=aug.device)
    info = torch.empty(B, dtype=torch.int32, device=aug.device)
    _lib.call("lc_batch_chol_solve", _p(aug), B, N, M, _p(linv), _p(h), _p(slot), _p(info),
              ctypes.byref(options) if options is not None else None, _s())
    return info


def batch_chol_inverse(aug, B, N, p, slot=None, options=None):
    """aug: (B, 2N, N) f64 systems with the identity as bottom block; p (B, N, N) f32 <- inverse of the top blocks."""
    linv = torch.empty((B, N // LC_NB, LC_NB, LC_NB), dtype=torch.float64, device=aug.device)
    info = torch.empty(B, dtype=torch.int32, device=aug.device)
    _lib.call("lc_batch_chol_inverse", _p(aug), B, N, _p(linv), _p(p), _p(slot), _p(info),
              ctypes.byref(options) if options is not None else None, _s())
    return info


def batch_eigh(a, max_sweeps=30, tol=1e-14):
    """Symmetric eigendecomposition of the (F, n, n) f64 systems ``a`` (n even; destroyed) by cyclic Jacobi:
    (eigenvalues (F, n), eigenvectors as ROWS (F, n, n), largest eigenvalue (F,), sweeps done)."""
    F, n, _ = a.shape
    _need(a, torch.float64, "batch_eigh")
    vt = torch.empty((F, n, n), dtype=torch.float64, device=a.device)
    lam = torch.empty((F, n), dtype=torch.float64, device=a.device)
    lmax = torch.empty(F, dtype=torch.float64, device=a.device)
    nbytes = int(_lib.load().lc_batch_eigh_work_bytes(F, n))
    work = torch.empty(nbytes, dtype=torch.uint8, device=a.device)
    sweeps = ctypes.c_int32(0)
    _lib.call("lc_batch_eigh_jacobi", _p(a), F, n, _p(vt), _p(lam), _p(lmax), _p(work), nbytes, int(max_sweeps), float(tol),
              ctypes.byref(sweeps), _s())
    return lam, vt, lmax, int(sweeps.value)


def batch_spectral_apply(lam, vt, r, a2, A, cutoff, rank_cap, h, slot=None):
    """h[slot[f A + a]] (M, n) f32 = r[f] V_kept diag(1 / (lam + a2[f A + a])) V_kept' (see lc_batch_spectral_apply);
    returns the number of eigenpairs kept per system (device int32)."""
    F, n = lam.shape
    M = r.shape[1]
    nbytes = int(_lib.load().lc_batch_spectral_work_bytes(F, A, n, M))
    work = torch.empty(nbytes, dtype=torch.uint8, device=lam.device)
    kept = torch.empty(F, dtype=torch.int32, device=lam.device)
    sl = None if slot is None else (ctypes.c_int32 * (F * A))(*[int(x) for x in slot])
    _lib.call("lc_batch_spectral_apply", _p(lam), _p(vt), F, n, _p(r), M, _p(a2), A, float(cutoff), _p(rank_cap), _p(work),
              nbytes, _p(h), sl, _p(kept), _s())
    return kept


def batch_series_hat(k, tr, va, F, N, M, scale, coef, aidx, A, terms, h):
    """coef: (S, terms) f64 polynomial coefficients of the S alphas (series.py); scale: (F) f64."""
    S = aidx.numel()
    work = torch.empty(F * N * N + terms * F * M * N, dtype=torch.float64, device=k.device)
    _lib.call("lc_batch_series_hat", _p(k), k.stride(0), _p(tr), _p(va), F, N, M, _p(scale), _p(coef), _p(aidx), S, A,
              terms, _p(work), _p(h), _s())


def batch_series_terms(k, tr, va, F, N, M, scale, terms, out, rowmap=None):
    """out: (F, rows_p, N) f32; ``rowmap`` (terms*M int32) places row i of term j, default j*M + i."""
    work = torch.empty(F * N * N + terms * F * M * N, dtype=torch.float64, device=k.device)
    _lib.call("lc_batch_series_terms", _p(k), k.stride(0), _p(tr), _p(va), F, N, M, _p(scale), terms, _p(work), _p(out),
              _p(rowmap), out.shape[1], _s())


def fisher_combine(p):
    """(k, V) f64 device p-values (NaN-free) -> (V,) f64 combined p-values."""
    k, V = p.shape
    out = torch.empty(V, dtype=torch.float64, device=p.device)
    _lib.call("lc_fisher_combine", _p(p), k, V, _p(out), _s())
    return out


def bh_fdr(p, alpha):
    """Benjamini-Hochberg on a (n,) f64 device vector -> (reject (n,) uint8, adjusted p (n,) f64), input order."""
    n = p.numel()
    nbytes = int(_lib.load().lc_bh_fdr_work_bytes(n))
    if nbytes < 0:
        raise ValueError("bh_fdr: bad length")
    work = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
    reject = torch.empty(n, dtype=torch.uint8, device=p.device)
    padj = torch.empty(n, dtype=torch.float64, device=p.device)
    _lib.call("lc_bh_fdr", _p(p), n, float(alpha), _p(reject), _p(padj), _p(work), nbytes, _s())
    return reject, padj


def bh_reject(p, alpha, want_status=False):
    """The Benjamini-Hochberg rejection mask alone ((n,) uint8, input order): no sort, no adjusted p-values.  The
    counting iteration is capped; ``want_status``: return (mask, status (1,) int32 device) without a host round trip --
    status != 0 means the mask was NOT written and the caller must take ``bh_fdr`` for this vector.  Otherwise the
    status is read here (a synchronisation) and the sort-based path is taken when needed."""
    n = p.numel()
    reject = torch.empty(n, dtype=torch.uint8, device=p.device)
    status = torch.empty(1, dtype=torch.int32, device=p.device)
    _lib.call("lc_bh_reject", _p(p), n, float(alpha), _p(reject), _p(status), _s())
    if want_status:
        return reject, status
    if int(status.cpu()[0]):
        reject = bh_fdr(p, alpha)[0]
    return reject


def gather_sub_f64(k, rows, cols, F, R, C, out):
    """out (F, R, C) f64 contiguous view = K[rows[f][i], cols[f][j]] (-1 -> 0)."""
    _lib.call("lc_gather_sub_f64", _p(k), k.stride(0), _p(rows), _p(cols), F, R, C, _p(out), _s())


def gather_sub_f32_strided(k, rows, cols, F, R, C, scale, out, s_f, s_r, s_c):
    """out[f * s_f + i * s_r + j * s_c] = K[rows[f][i], cols[f][j]] / scale[f] as f32."""
    _lib.call("lc_gather_sub_f32_strided", _p(k), k.stride(0), _p(rows), _p(cols), F, R, C, _p(scale), _p(out), s_f, s_r, s_c,
              _s())


def series_place(q, N, F, ldq, M, rowmap, P, rows_p):
    _lib.call("lc_series_place", _p(q), N, F, ldq, M, _p(rowmap), _p(P), rows_p, _s())


def scale_cast_f64_f32(src, divisor, dst):
    _lib.call("lc_scale_cast_f64_f32", _p(src), _p(divisor), _p(dst), src.numel(), _s())
    return dst


def combine_terms(terms, coef, out):
    """out = sum_j coef[j] * terms[j] (<= 4 contiguous f32 tensors of out's size), left to right in fp32."""
    n = len(terms)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in terms])
    cf = (ctypes.c_float * n)(*[float(c) for c in coef])
    _lib.call("lc_combine_terms_f32", ptrs, cf, n, _p(out), out.numel(), _s())
    return out


def combine_many(terms, coef, out):
    """out = sum_j coef[j] * terms[j] for ANY number of contiguous tensors of out's size and dtype (f32 / f64), left to
    right: lc_combine_terms_f32 / _f64 in chunks of four (the running sum is the first term of the next chunk)."""
    f64 = out.dtype == torch.float64
    name, ctype = ("lc_combine_terms_f64", ctypes.c_double) if f64 else ("lc_combine_terms_f32", ctypes.c_float)
    terms, coef = list(terms), [float(c) for c in coef]
    first = True
    while terms:
        take = 4 if first else 3
        ts, cs = terms[:take], coef[:take]
        terms, coef = terms[take:], coef[take:]
        if not first:
            ts, cs = [out] + ts, [1.0] + cs
        n = len(ts)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        _lib.call(name, ptrs, (ctype * n)(*cs), n, _p(out), out.numel(), _s())
        first = False
    return out


def combine_colmax(terms, coef, out, cols, want_scales_for=None):
    """out = sum_j coef[j] * terms[j] over (rows, ld) float32 matrices (contiguous, ld = out.stride(0)), any number of
    terms, left to right as ``combine_many``; with ``want_scales_for`` = V also the fp16 column scales of the result
    (``col_scales_f16``'s, (2 V,) float32) from the column maxima the last pass takes while it writes."""
    terms, coef = list(terms), [float(c) for c in coef]
    rows, ld = out.shape[0], out.stride(0)
    colmax = zeros(cols, torch.int32, out.device) if want_scales_for else None
    first = True
    while terms:
        take = 4 if first else 3
        ts, cs = terms[:take], coef[:take]
        terms, coef = terms[take:], coef[take:]
        if not first:
            ts, cs = [out] + ts, [1.0] + cs
        n = len(ts)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
        _lib.call("lc_combine_terms_colmax_f32", ptrs, (ctypes.c_float * n)(*cs), n, _p(out), ld, rows, cols,
                  _p(colmax) if not terms else None, _s())
        first = False
    if not want_scales_for:
        return out, None
    cs_out = torch.empty(2 * want_scales_for, dtype=torch.float32, device=out.device)
    _lib.call("lc_col_scales_from_max", _p(colmax), want_scales_for, _p(cs_out), _s())
    return out, cs_out


def gather_sub_f32(k, rows, cols, F, R, C, scale, out):
    _lib.call("lc_gather_sub_f32", _p(k), k.stride(0), _p(rows), _p(cols), F, R, C, _p(scale), _p(out), _s())


def series_scores(t, ldt, terms, M, n_val, V, yv, ystat, coef, aidx, scores, accumulate, rowmap=None):
    _lib.call("lc_series_scores", _p(t), ldt, terms, M, n_val, V, _p(yv), _p(ystat), _p(coef), _p(aidx), aidx.numel(),
              _p(rowmap), _p(scores), int(bool(accumulate)), _s())


def transpose_rows(x, tr, N, p, out):
    _lib.call("lc_transpose_rows_f64", _p(x), x.stride(0), _p(tr), N, p, _p(out), _s())


# ------------------------------------------------------------------ V-wide contractions
def val_stats(y, V, va, M, n_val, ystat, yblk, yv):
    _lib.call("lc_val_stats", _p(y), y.stride(0), V, _p(va), M, n_val, _p(ystat), _p(yblk), _p(yv), _s())


def val_stats_folds(y, V, va, F, M, n_vals, ystat, yblk, yv, live=None):
    """All F inner folds of an outer fold in one launch: va (F, M), outputs (F, 3, V) / (F, M/32, V) / (F, M, V)."""
    import ctypes as ct
    nv = (ct.c_int32 * F)(*[int(n) for n in n_vals])
    _lib.call("lc_val_stats_folds", _p(y), y.stride(0), V, _p(va), F, M, nv, _p(ystat), _p(yblk), _p(yv), _p(live), _s())


def alpha_sweep_scores(h, A, M, N, y, V, tr, yv, n_val, ystat, yblk, mode, part, scores, accumulate):
    _lib.call("lc_alpha_sweep_scores", _p(h), A, M, N, _p(y), y.stride(0), V, _p(tr), _p(yv), n_val, _p(ystat),
              _p(yblk), mode, _p(part), _p(scores), int(bool(accumulate)), _s())


def split_rows_f16(h, rows, K, tiled, rowscale_inv):
    _lib.call("lc_split_rows_f16", _p(h), h.stride(0), rows, K, _p(tiled), _p(rowscale_inv), _s())


def split_rows_f16_alphas(h, groups, A, M, K, tiled, rowscale_inv):
    """The A image of the fused sweep (alpha_sweep_scores_f16x3): per group A hat matrices of M rows each, 32-row blocks in
    (validation block, alpha) order -- see lc_split_rows_f16_alphas."""
    _lib.call("lc_split_rows_f16_alphas", _p(h), h.stride(0), groups, A, M, K, _p(tiled), _p(rowscale_inv), _s())


def split_rows_f16_alphas_sel(h, groups, A_src, A, M, K, tiled, rowscale_inv):
    """split_rows_f16_alphas of the first ``A`` of the ``A_src`` hat matrices each group holds (lc_split_rows_f16_alphas_sel)."""
    _lib.call("lc_split_rows_f16_alphas_sel", _p(h), h.stride(0), groups, A_src, A, M, K, _p(tiled), _p(rowscale_inv), _s())


def split_rows_f16_groups(h, groups, rows, K, tiled, rowscale_inv):
    """``groups`` consecutive row blocks of ``rows`` rows of h, each padded to whole 256-row tiles in the image."""
    _lib.call("lc_split_rows_f16_groups", _p(h), h.stride(0), groups, rows, K, _p(tiled), _p(rowscale_inv), _s())


def mean_operator_image(mats, maps, scale, rows, K, tiled, rowscale_inv):
    """The tiled fp16 image of  scale * sum_f scatter_f(mats[f][:rows])  -- the mean of the folds' refit operators, fold f's
    columns sent to the target rows ``maps[f]`` lists ((K,) int32 device: column of mats[f] per target row, -1 none) -- and
    its row scales (lc_mean_operator_image_f16)."""
    n = len(mats)
    m = (ctypes.c_void_p * n)(*[x.data_ptr() for x in mats])
    ld = (ctypes.c_int64 * n)(*[int(x.stride(0)) for x in mats])
    mp = (ctypes.c_void_p * n)(*[x.data_ptr() for x in maps])
    _lib.call("lc_mean_operator_image_f16", m, ld, mp, n, float(scale), rows, K, _p(tiled), _p(rowscale_inv), _s())


def mean_operator_images(mats_per_image, slots, maps, scale, rows, K, tiled, rowscale_inv):
    """The images of several alpha tuples in one launch (lc_mean_operator_images_f16): ``mats_per_image[i]`` = the folds'
    operators of image i (all folds' operators of one fold share their row stride), written at slot ``slots[i]`` of ``tiled`` /
    ``rowscale_inv`` (pad256(rows) * K * 2 halves / pad256(rows) floats per slot).  Bit for bit what mean_operator_image
    writes, image by image."""
    n_img = len(mats_per_image)
    if n_img == 0:
        return
    n = len(maps)
    ld = [int(x.stride(0)) for x in mats_per_image[0]]
    table = np.empty((n_img, n + 1), dtype=np.int64)
    for i, (mats, slot) in enumerate(zip(mats_per_image, slots)):
        if len(mats) != n or any(int(x.stride(0)) != l for x, l in zip(mats, ld)):
            raise ValueError("mean_operator_images: every image takes one operator per fold, with the fold's row stride")
        table[i, :n] = [x.data_ptr() for x in mats]
        table[i, n] = int(slot)
    d_table = upload(table, tiled.device)
    c_ld = (ctypes.c_int64 * n)(*ld)
    mp = (ctypes.c_void_p * n)(*[x.data_ptr() for x in maps])
    for i0 in range(0, n_img, 65535):
        k = min(65535, n_img - i0)
        _lib.call("lc_mean_operator_images_f16", ctypes.c_void_p(d_table.data_ptr() + i0 * (n + 1) * 8), k, c_ld, mp, n,
                  float(scale), rows, K, _p(tiled), _p(rowscale_inv), _s())


def col_scales_f16(y, T, V, want_flag=True, colflags=None, live=None):
    """(cs, flag): cs[:V] = 2^-e, cs[V:] = 2^e per column; flag (device int32) != 0 when some column's
    dynamic range is too wide for the fp16 hi/lo split (``want_flag=False``: the scales alone, one pass over y).
    ``colflags``: (V,) uint8 device vector that receives WHICH columns raised the flag."""
    cs = torch.empty(2 * V, dtype=torch.float32, device=y.device)
    flag = zeros(1, torch.int32, y.device) if want_flag else None
    if colflags is not None or live is not None:
        _lib.call("lc_col_scales_f16_flags", _p(y), y.stride(0), T, V, _p(cs), _p(flag), _p(colflags), _p(live), _s())
    else:
        _lib.call("lc_col_scales_f16", _p(y), y.stride(0), T, V, _p(cs), _p(flag), _s())
    return cs, flag


def scatter_cols(src, n_rows, cols, n_cols, dst):
    """dst[r, cols[j]] = src[r, j] (overwrite; cols[j] < 0 skipped): 2-D f32 / f64 device matrices (or views with a row
    stride), ``cols`` an int32 device vector of distinct indices."""
    if src.dtype != dst.dtype or src.dtype not in (torch.float32, torch.float64):
        raise ValueError("scatter_cols: f32 or f64 matrices of the same type")
    _lib.call("lc_scatter_cols", _p(src), src.stride(0) if src.dim() > 1 else max(int(n_cols), 1), n_rows, src.element_size(),
              _p(cols), n_cols, _p(dst), dst.stride(0) if dst.dim() > 1 else 0, _s())
    return dst


GEMV_MAX_COLS = 8


def gemv_cols(a, M, K, y, rows, ns, out, sel=None, want=0):
    """out[m, j] = sum_k a[m, k] * y[rows[k], j] for the ns <= GEMV_MAX_COLS first columns of ``y`` (those with sel[j] ==
    want when ``sel`` is given; the others are left alone): f32 in, fp64 accumulation, f32 out (lc_gemv_cols_f32).
    ``a``: (M, >= K) f32 rows of stride a.stride(-2); ``rows``: int32 device list (K) or None."""
    _lib.call("lc_gemv_cols_f32", _p(a), a.stride(-2), M, K, _p(y), y.stride(0), _p(rows), _p(sel), int(ns), int(want),
              _p(out), out.stride(0), _s())
    return out


def split_cols_f16(y, V, rows, K, cscale, tiled, live=None):
    _lib.call("lc_split_cols_f16", _p(y), y.stride(0), V, _p(rows), K, _p(cscale), _p(tiled), _p(live), _s())


def permute_cols_f16(tiled, perm, Vs, K, out):
    """out = the tiled fp16 image ``tiled`` (K rows) with column j taken from column perm[j] (-1: zeros), j < Vs."""
    _lib.call("lc_permute_cols_f16", _p(tiled), _p(perm), Vs, K, _p(out), _s())
    return out


def alpha_sweep_scores_f16x3(ht, rowscale_inv, A, M, N, yt, cscale_inv, yv, V, n_val, ystat, yblk, mode, part, scores,
                             accumulate, bview=(0, 0, 0), terms=3, live=None):
    """One inner fold (F = 1 of lc_alpha_sweep_scores_f16x3_folds).  ``bview`` = (rows of the tiled image yt, first row
    of the skipped block, its length); (0, 0, 0): yt holds exactly the N contracted rows."""
    alpha_sweep_scores_f16x3_folds(ht, rowscale_inv, A, M, N, yt, cscale_inv, yv, V, [n_val], ystat, yblk, mode, part, scores,
                                   accumulate, [bview], terms=terms, live=live)


def gemm_grouped_f16x3(at, rowscale_inv, Mrows, bt, cscale_inv, c, ldc, Ncols, K, group_tiles, slab_light=None,
                       bview=(0, 0, 0)):
    G = len(group_tiles) - 1
    if G > GROUP_RANGE:
        # more column groups than one launch carries (a refit over more than 64 distinct alphas): one launch per range
        # of 64 groups, on views of the operands (group g's A image / row scales and column tile t's B image, column
        # scales and output columns are contiguous blocks)
        if slab_light is not None or tuple(bview) != (0, 0, 0):
            raise ValueError("gemm_grouped_f16x3: more than 64 groups only without slab flags / B views")
        a_stride = pad_to(Mrows, 256) * K * 2
        for g0 in range(0, G, GROUP_RANGE):
            g1 = min(G, g0 + GROUP_RANGE)
            t0, t1 = int(group_tiles[g0]), int(group_tiles[g1])
            if t1 == t0:
                continue
            gemm_grouped_f16x3(at[g0 * a_stride:], rowscale_inv[g0 * pad_to(Mrows, 256):], Mrows, bt[t0 * 256 * K * 2:],
                               cscale_inv[t0 * 256:], c[:, t0 * 256:], ldc, (t1 - t0) * 256, K,
                               [int(t) - t0 for t in group_tiles[g0:g1 + 1]])
        return
    arr = (ctypes.c_int32 * (G + 1))(*[int(t) for t in group_tiles])
    _lib.call("lc_gemm_grouped_f16x3", _p(at), _p(rowscale_inv), Mrows, _p(bt), _p(cscale_inv), _p(c), ldc, Ncols, K,
              arr, G, _p(slab_light), *bview, _s())


def gemm_grouped_f16x3_pearson(at, rowscale_inv, Mrows, bt, cscale_inv, Ncols, K, group_tiles, y, y_rows, y_cols, r_out):
    """Pearson r per column between the rows of the grouped product (never stored) and the targets
    ``y[y_rows[i], y_cols[column]]`` (None lists: row i / the column itself), into ``r_out`` (Ncols,) float64: the test
    rows of the refit (lc_gemm_grouped_f16x3_pearson).  More than 64 column groups: one launch per range of 64."""
    G = len(group_tiles) - 1
    slabs = -(-Mrows // 128)
    a_stride = pad_to(Mrows, 256) * K * 2
    for g0 in range(0, G, GROUP_RANGE):
        g1 = min(G, g0 + GROUP_RANGE)
        t0, t1 = int(group_tiles[g0]), int(group_tiles[g1])
        if t1 == t0:
            continue
        n = (t1 - t0) * 256
        part = torch.empty(slabs * 6 * n, dtype=torch.float64, device=r_out.device)
        arr = (ctypes.c_int32 * (g1 - g0 + 1))(*[int(t) - t0 for t in group_tiles[g0:g1 + 1]])
        y_c = y if y_cols is not None else y[:, t0 * 256:]
        _lib.call("lc_gemm_grouped_f16x3_pearson", _p(at[g0 * a_stride:]), _p(rowscale_inv[g0 * pad_to(Mrows, 256):]), Mrows,
                  _p(bt[t0 * 256 * K * 2:]), _p(cscale_inv[t0 * 256:]), n, K, arr, g1 - g0, _p(y_c), y.stride(0), _p(y_rows),
                  _p(None if y_cols is None else y_cols[t0 * 256:]), _p(part), _p(r_out[t0 * 256:]), _s())
    return r_out


def series_sweep_scores_f16x3(pt, rowscale_inv, M, n_val, K, yt, cscale_inv, Ncols, yv, V, ystat, yblk, coef, aidx, part,
                              scores, accumulate, bview=(0, 0, 0), terms=3, live=None):
    """Series contraction with the moments epilogue + the scores of the series alphas, one inner fold (F = 1 of
    lc_series_sweep_scores_f16x3_folds)."""
    series_sweep_scores_f16x3_folds(pt, rowscale_inv, M, [n_val], K, yt, cscale_inv, Ncols, yv, V, ystat, yblk, coef, aidx,
                                    part, scores, accumulate, [bview], terms=terms, live=live)


def _fold_arrays(n_vals, views):
    F = len(n_vals)
    nv = (ctypes.c_int32 * F)(*[int(n) for n in n_vals])
    b_rows = int(views[0][0]) if views else 0
    g0 = (ctypes.c_int64 * F)(*[int(v[1]) for v in views])
    gl = (ctypes.c_int64 * F)(*[int(v[2]) for v in views])
    return F, nv, b_rows, g0, gl


def alpha_sweep_scores_f16x3_folds(ht, rowscale_inv, A, M, N, yt, cscale_inv, yv, V, n_vals, ystat, yblk, mode, part, scores,
                                   accumulate, views, terms=3, live=None):
    """All inner folds in one launch (see lc_alpha_sweep_scores_f16x3_folds); views: per fold (b_rows, gap0, gap rows).
    ``terms``: 3 = fp32-level products, 1 = the screening pass (hi planes only); ``live``: device int32 -- only the
    column tiles below that many columns do any work."""
    F, nv, b_rows, g0, gl = _fold_arrays(n_vals, views)
    _lib.call("lc_alpha_sweep_scores_f16x3_folds", _p(ht), _p(rowscale_inv), F, A, M, N, _p(yt), _p(cscale_inv), _p(yv), V,
              nv, _p(ystat), _p(yblk), mode, _p(part), _p(scores), 2 if accumulate == 2 else int(bool(accumulate)), b_rows, g0, gl,
              int(terms), _p(live), _s())


def series_sweep_scores_f16x3_folds(pt, rowscale_inv, M, n_vals, K, yt, cscale_inv, Ncols, yv, V, ystat, yblk, coef, aidx,
                                    part, scores, accumulate, views, terms=3, live=None):
    F, nv, b_rows, g0, gl = _fold_arrays(n_vals, views)
    _lib.call("lc_series_sweep_scores_f16x3_folds", _p(pt), _p(rowscale_inv), F, M, nv, K, _p(yt), _p(cscale_inv), Ncols,
              _p(yv), V, _p(ystat), _p(yblk), _p(coef), _p(aidx), aidx.numel(), _p(part), _p(scores),
              2 if accumulate == 2 else int(bool(accumulate)), b_rows, g0, gl, int(terms), _p(live), _s())


def alpha_sweep_finalize_folds(part, ystat, yblk, A, M, n_vals, V, mode, scores, accumulate=False, live=None):
    """Scores of F folds whose fused contractions were launched with accumulate=2 (lc_alpha_sweep_finalize_folds)."""
    F = len(n_vals)
    nv = (ctypes.c_int32 * F)(*[int(n) for n in n_vals])
    _lib.call("lc_alpha_sweep_finalize_folds", _p(part), _p(ystat), _p(yblk), F, A, M, nv, V, mode, _p(scores),
              int(bool(accumulate)), _p(live), _s())


def series_sweep_finalize_folds(part, ystat, yblk, M, n_vals, V, coef, aidx, scores, accumulate=False, live=None):
    """... and of the series-moments contractions (lc_series_sweep_finalize_folds)."""
    F = len(n_vals)
    nv = (ctypes.c_int32 * F)(*[int(n) for n in n_vals])
    _lib.call("lc_series_sweep_finalize_folds", _p(part), _p(ystat), _p(yblk), F, M, nv, V, _p(coef), _p(aidx), aidx.numel(),
              _p(scores), int(bool(accumulate)), _p(live), _s())


def undecided_cols(scores, A, V, tau_sum, ystat, cap):
    """(list (cap,) int32 of the undecided columns of the screening pass' score sums, ascending, -1 behind them; count (3,)
    int32: columns in the list / undecided columns found / 1 when the list does not hold them all) -- lc_undecided_cols.  ``scores``: (A, >= V) f32 of row stride
    scores.stride(0); ``ystat``: (3, ld) f32 validation statistics of one inner fold, or None."""
    dev = scores.device
    flags = torch.empty(V, dtype=torch.uint8, device=dev)
    blocks = torch.empty((V + 255) // 256, dtype=torch.int32, device=dev)
    lst = torch.empty(cap, dtype=torch.int32, device=dev)
    count = torch.empty(3, dtype=torch.int32, device=dev)
    _lib.call("lc_undecided_cols", _p(scores), A, scores.stride(0), V, float(tau_sum), _p(ystat),
              ystat.stride(0) if ystat is not None else 0, _p(flags), _p(blocks), _p(lst), int(cap), _p(count), _s())
    return lst, count


def kappa_sums(ystat, V):
    """(2,) f64 device: sum and sum of squares over V voxels of kappa = rms / std of the validation rows (lc_kappa_sums)."""
    out = torch.empty(2, dtype=torch.float64, device=ystat.device)
    _lib.call("lc_kappa_sums", _p(ystat), ystat.stride(0), V, _p(out), _s())
    return out


def select_alpha(scores, A, V, want_best=True, want_rowsum=False):
    best = torch.empty(V, dtype=torch.int32, device=scores.device) if want_best else None
    rowsum = torch.empty(A, dtype=torch.float64, device=scores.device) if want_rowsum else None
    _lib.call("lc_select_alpha", _p(scores), A, V, _p(best), _p(rowsum), _s())
    return best, rowsum


def accumulate_f64(x, acc):
    _lib.call("lc_accumulate_f64", _p(x), _p(acc), x.numel(), _s())
    return acc


def fill_argmax(rowsum, A, best, V):
    _lib.call("lc_fill_argmax", _p(rowsum), A, _p(best), V, _s())
    return best


def fold_pack(r_s, p_s, perm, Vs, best, V, info_a, info_b, out, col0=0, clear=True):
    """out: (4, ld) f64 device block (see lc_fold_pack_at): this range's V voxels land in columns [col0, col0 + V)."""
    _lib.call("lc_fold_pack_at", _p(r_s), _p(p_s), _p(perm), Vs, _p(best), V, _p(info_a), info_a.numel(), _p(info_b),
              info_b.numel(), _p(out), out.stride(0), int(col0), int(bool(clear)), _s())
    return out


def fold_unpack(src, world, ld, lo, w_max, r, p, idx, p_clean, bad):
    _lib.call("lc_fold_unpack", _p(src), world, ld, _p(lo), w_max, _p(r), _p(p), _p(idx), _p(p_clean), _p(bad), _s())


GROUP_RANGE = 64                    # alpha groups one grouping launch / one grouped GEMM launch carries


def group_by_alpha(best, V, A, pad):
    """Counting sort of the voxels by alpha index.  Returns (perm, count (2, A)): for A <= 64 ``perm`` is the one device
    vector of lc_group_by_alpha; for a larger grid (the reference groups by any number of distinct alphas,
    ridge_regression.py:46-50) a LIST with one such vector per range of 64 alphas (lc_group_by_alpha_range) -- the caller
    joins them once the group sizes are on the host (join_group_ranges)."""
    count = torch.empty((2, A), dtype=torch.int32, device=best.device)       # row 0: counts; row 1: a copy (callers
    if A <= GROUP_RANGE:                                                      # all-reduce it over shards)
        perm = filled((V + A * pad,), torch.int32, best.device, 0xFF)         # -1 everywhere
        _lib.call("lc_group_by_alpha", _p(best), V, A, pad, _p(perm), _p(count), _s())
    else:
        perm = []
        for a0 in range(0, A, GROUP_RANGE):
            n = min(GROUP_RANGE, A - a0)
            part = filled((V + n * pad,), torch.int32, best.device, 0xFF)
            _lib.call("lc_group_by_alpha_range", _p(best), V, a0, n, pad, _p(part), _p(count[0, a0:]), _s())
            perm.append(part)
    count[1].copy_(count[0])
    return perm, count


def join_group_ranges(perms, counts, pad):
    """One alpha-sorted voxel list from the per-range lists of group_by_alpha (A > 64): range k's groups, each padded to
    ``pad`` columns, follow range k-1's.  ``counts``: the (A,) group sizes on the host."""
    lens = []
    for k in range(len(perms)):
        c = counts[k * GROUP_RANGE:(k + 1) * GROUP_RANGE]
        lens.append(int(sum((int(x) + pad - 1) // pad * pad for x in c)))
    out = filled((max(sum(lens), 1),), torch.int32, perms[0].device, 0xFF)
    o = 0
    for part, n in zip(perms, lens):
        if n:
            out[o:o + n].copy_(part[:n])
        o += n
    return out


def gemm_grouped(a, lda, a_group_stride, b, ldb, brows, c, ldc, Mrows, Ncols, K, group_tiles):
    G = len(group_tiles) - 1
    if G > GROUP_RANGE:                                 # (see gemm_grouped_f16x3: one launch per range of 64 groups)
        for g0 in range(0, G, GROUP_RANGE):
            g1 = min(G, g0 + GROUP_RANGE)
            t0, t1 = int(group_tiles[g0]), int(group_tiles[g1])
            if t1 > t0:
                gemm_grouped(a.reshape(-1)[g0 * a_group_stride:], lda, a_group_stride, b[:, t0 * COL_TILE:], ldb, brows,
                             c[:, t0 * COL_TILE:], ldc, Mrows, (t1 - t0) * COL_TILE, K,
                             [int(t) - t0 for t in group_tiles[g0:g1 + 1]])
        return
    arr = (ctypes.c_int32 * (G + 1))(*[int(t) for t in group_tiles])
    _lib.call("lc_gemm_grouped_f32", _p(a), lda, a_group_stride, _p(b), ldb, _p(brows), _p(c), ldc, Mrows, Ncols, K,
              arr, G, _s())