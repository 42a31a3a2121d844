"""CPU: the C-ABI library loads and exports every symbol include/ssak_hip.h declares (no compute)."""
import ctypes
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    lib_path = os.path.join(ROOT, "ssak_amd", "lib", "libssak_hip.so")
    assert os.path.exists(lib_path), "build first: make (or __graft_entry__.build())"
    lib = ctypes.CDLL(lib_path)
    hdr = open(os.path.join(ROOT, "include", "ssak_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(ssak_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 8
    for n in sorted(names):
        assert hasattr(lib, n), n
    lib.ssak_version.restype = ctypes.c_int
    assert lib.ssak_version() >= 100


def test_binding_rejects_without_gpu_compute():
    import ssak_amd.hip as h
    # argument validation happens on the host, before any launch
    d = h.GemmDesc(0, 8, 8, 0, 0, 8, 8, 8, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0, 1, 0, 1, 0.0, 0, 0, 0, 0, 0)
    rc = h.lib.ssak_gemm_bf16(ctypes.byref(d), None, None, None, None, None, None, None, 0, None)
    assert rc == h.SSAK_ERR_INVALID and b"null" in h.lib.ssak_last_error()


def test_read_ranges_reads_and_reports(tmp_path):
    """ssak_read_ranges (host: the file reads of an ingest batch by native threads): bytes land where asked for any thread count,
    empty ranges are skipped, a missing file and a range past the end of a file are SSAK_ERR_INVALID with the path in the message."""
    import numpy as np
    import ssak_amd.hip as h
    rng = np.random.default_rng(0)
    blobs, paths = [], []
    for i in range(11):
        b = rng.integers(0, 256, 1000 + 37 * i, dtype=np.uint8)
        p = tmp_path / f"f{i}.bin"
        p.write_bytes(b.tobytes())
        blobs.append(b)
        paths.append(os.fsencode(str(p)))
    n = len(paths)
    off = [7 * i for i in range(n)]
    size = [len(blobs[i]) - off[i] - (i % 3) for i in range(n)]
    size[4] = 0
    pos = np.cumsum([0] + size[:-1])
    for threads in (1, 3, 8, 64):
        buf = np.full(sum(size) + 16, 0xEE, dtype=np.uint8)
        args = ((ctypes.c_char_p * n)(*paths), (ctypes.c_int64 * n)(*off), (ctypes.c_int64 * n)(*size),
                (ctypes.c_void_p * n)(*[buf.ctypes.data + int(q) for q in pos]))
        assert h.lib.ssak_read_ranges(*args, n, threads) == h.SSAK_OK
        for i in range(n):
            assert (buf[pos[i]:pos[i] + size[i]] == blobs[i][off[i]:off[i] + size[i]]).all(), (threads, i)
        assert (buf[sum(size):] == 0xEE).all()
    buf = np.zeros(64, dtype=np.uint8)
    one = lambda path, o, s: h.lib.ssak_read_ranges((ctypes.c_char_p * 1)(path), (ctypes.c_int64 * 1)(o), (ctypes.c_int64 * 1)(s),
                                                    (ctypes.c_void_p * 1)(buf.ctypes.data), 1, 4)
    assert one(os.fsencode(str(tmp_path / "nope.bin")), 0, 8) == h.SSAK_ERR_INVALID and b"nope.bin" in h.lib.ssak_last_error()
    assert one(paths[0], 990, 64) == h.SSAK_ERR_INVALID and b"short read" in h.lib.ssak_last_error()
    assert h.lib.ssak_read_ranges(None, None, None, None, 0, 4) == h.SSAK_OK


def test_read_ranges_pool_is_reused_and_takes_concurrent_callers(tmp_path):
    """The reader threads are a process-wide pool that outlives a call: hundreds of calls with changing thread counts, from several
    Python threads at once (two ingests take turns), and in a forked child (which starts over with an empty pool) all read the
    right bytes; ssak_drop_file_cache leaves the files readable."""
    import threading

    import numpy as np
    import ssak_amd.hip as h
    rng = np.random.default_rng(1)
    blobs, paths = [], []
    for i in range(24):
        b = rng.integers(0, 256, 5000 + 11 * i, dtype=np.uint8)
        p = tmp_path / f"g{i}.bin"
        p.write_bytes(b.tobytes())
        blobs.append(b)
        paths.append(os.fsencode(str(p)))

    def once(threads, sel):
        n = len(sel)
        size = [len(blobs[i]) for i in sel]
        pos = np.cumsum([0] + size[:-1])
        buf = np.zeros(sum(size), dtype=np.uint8)
        rc = h.lib.ssak_read_ranges((ctypes.c_char_p * n)(*[paths[i] for i in sel]), (ctypes.c_int64 * n)(*([0] * n)), (ctypes.c_int64 * n)(*size),
                                    (ctypes.c_void_p * n)(*[buf.ctypes.data + int(q) for q in pos]), n, threads)
        return rc == h.SSAK_OK and all((buf[pos[k]:pos[k] + size[k]] == blobs[i]).all() for k, i in enumerate(sel))

    ok = []

    def caller(seed):
        r = np.random.default_rng(seed)
        for _ in range(100):
            sel = r.choice(24, int(r.integers(1, 24)), replace=False).tolist()
            ok.append(once(int(r.integers(1, 17)), sel))

    ts = [threading.Thread(target=caller, args=(s,)) for s in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert len(ok) == 400 and all(ok)
    n = len(paths)
    assert h.lib.ssak_drop_file_cache((ctypes.c_char_p * n)(*paths), n) == 0
    assert once(8, list(range(24)))
    pid = os.fork()
    if pid == 0:  # the child has none of the pool's threads
        os._exit(0 if once(8, list(range(24))) and once(3, [5, 6, 7]) else 1)
    assert os.waitpid(pid, 0)[1] == 0


def test_no_scratch_inside_the_matrix_loops():
    """Register spills where they would hurt.  The persistent GEMMs carry a few spilled registers in their per-tile prologue /
    epilogue blocks (harmless, outside the K loop); one more live value in the loop and `-Rpass-analysis` would look the same.
    tools/loop_scratch.py disassembles the device code of the BUILT library and finds, per gemm_p4 / gemm_p8 / attention
    instantiation, the innermost loops that contain MFMAs: none of them may hold a scratch instruction."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import loop_scratch
    rows = loop_scratch.check()
    assert len(rows) >= 40 and all(nl >= 1 for _, _, _, nl in rows), "expected every hot instantiation with its K loop"
    bad = [(n, inside) for n, _, inside, _ in rows if inside]
    assert not bad, f"scratch instructions inside K loops: {bad}"


def test_base_gradient_buckets_cover_the_trainable_range():
    """The gradient ranges the backward announces for the data-parallel exchange, from the configuration alone
    (ssak_w2v2_grad_ranges: host arithmetic, no GPU): for wav2vec2-base with the frozen feature encoder they are disjoint,
    16-byte aligned and cover [0, 90 195 872) -- the 360.8 MB fp32 all-reduce payload of SURVEY.md section 8e -- as the head
    matrix, twelve 28.3 MB layer buckets (last layer first), the leading small matrices and the vector region."""
    from ssak_amd.config import Wav2Vec2Config
    from ssak_amd.model import Wav2Vec2ForCTC
    r = Wav2Vec2ForCTC.grad_ranges(Wav2Vec2Config())
    assert len(r) == 1 + 12 + 2
    assert all(off % 4 == 0 and cnt > 0 for off, cnt in r)  # 4 floats = 16 bytes
    spans = sorted(r)
    assert spans[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(spans, spans[1:]))
    assert spans[-1][0] + spans[-1][1] == 90_195_872
    layer = 4 * 768 * 768 + 2 * 768 * 3072
    assert [c for _, c in r[1:13]] == [layer] * 12 and r[0][1] == 32 * 768
    assert [o for o, _ in r[1:13]] == sorted((o for o, _ in r[1:13]), reverse=True)  # announced from the last layer down
    # the trained feature encoder (--no_freeze) extends the last range to all 94 396 320 parameters
    r2 = Wav2Vec2ForCTC.grad_ranges(Wav2Vec2Config(), freeze_feature_encoder=False)
    assert sum(c for _, c in r2) == 94_396_320
    # XLSR-large: 24 layer buckets of 50.3 MB
    x = Wav2Vec2ForCTC.grad_ranges(Wav2Vec2Config(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                                                   feat_extract_norm="layer", conv_bias=True, do_stable_layer_norm=True))
    assert len(x) == 27 and sorted(x)[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(sorted(x), sorted(x)[1:]))
