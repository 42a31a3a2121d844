"""ctypes binding of ``libssak_hip.so`` (C ABI: ``include/ssak_hip.h``).

torch is used here only as plumbing: device buffers (``tensor.data_ptr()``) and the current HIP stream.
Every wrapper raises ``RuntimeError`` when the library reports a launch failure and ``ValueError`` when
it rejects the arguments -- the exception types the reference's Python call sites would see.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_DEFAULT_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libssak_hip.so")
_LIB_PATH = os.environ.get("SSAK_HIP_LIB") or _DEFAULT_LIB  # (the override names another build OF THE SAME ABI: A/B runs, instrumented builds)
ABI_VERSION = 510  # ssak_version() of the library this binding's struct layouts and signatures were written for


class GemmDesc(C.Structure):
    _fields_ = [("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("a_kmajor", C.c_int), ("b_kmajor", C.c_int),
                ("lda", C.c_long), ("ldb", C.c_long), ("ldc", C.c_long), ("nb1", C.c_int), ("nb2", C.c_int),
                ("sa1", C.c_long), ("sa2", C.c_long), ("sb1", C.c_long), ("sb2", C.c_long), ("sc1", C.c_long),
                ("sc2", C.c_long), ("alpha", C.c_float), ("epilogue", C.c_int), ("out_f32", C.c_int),
                ("accumulate", C.c_int), ("split_k", C.c_int), ("drop_p", C.c_float), ("drop_stream", C.c_uint32),
                ("drop_seed", C.c_uint64), ("bias_s2", C.c_long), ("pads_are_zero", C.c_int), ("colsum", C.c_int),
                ("dynamic_tiles", C.c_int), ("b_fragments", C.c_void_p), ("plan_tile", C.c_int)]


class W2V2Config(C.Structure):
    _fields_ = [("vocab_size", C.c_int), ("hidden_size", C.c_int), ("num_layers", C.c_int), ("num_heads", C.c_int),
                ("intermediate_size", C.c_int), ("num_conv_layers", C.c_int), ("conv_dim", C.c_int * 8),
                ("conv_kernel", C.c_int * 8), ("conv_stride", C.c_int * 8), ("conv_bias", C.c_int),
                ("feat_extract_norm", C.c_int), ("do_stable_layer_norm", C.c_int),
                ("num_conv_pos_embeddings", C.c_int), ("num_conv_pos_embedding_groups", C.c_int),
                ("layer_norm_eps", C.c_float), ("attention_dropout", C.c_float), ("hidden_dropout", C.c_float),
                ("activation_dropout", C.c_float), ("feat_proj_dropout", C.c_float), ("final_dropout", C.c_float),
                ("freeze_feature_encoder", C.c_int), ("arch", C.c_int), ("num_mel_bins", C.c_int),
                ("max_source_positions", C.c_int), ("exact", C.c_int)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 112), ("launches", C.c_long), ("total_ms", C.c_double), ("total_flops", C.c_double),
                ("bound", C.c_int)]


GRAD_READY_FN = C.CFUNCTYPE(None, C.c_long, C.c_long, C.c_void_p)


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(f"{_LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()); "
                          "ssak_amd has no CPU fallback")
    lib = C.CDLL(_LIB_PATH)
    vp, i32, f32, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    sig = {
        "ssak_version": (i32, []),
        "ssak_last_error": (C.c_char_p, []),
        "ssak_wave_normalize_workspace_bytes": (sz, [i32, i32]),
        "ssak_wave_normalize": (i32, [vp, vp, i32, i32, vp, vp, vp, sz, vp]),
        "ssak_logmel_table_floats": (sz, []),
        "ssak_logmel_init_tables": (i32, [vp]),
        "ssak_logmel_workspace_bytes": (sz, [i32, i32]),
        "ssak_logmel_whisper": (i32, [vp, vp, i32, i32, i32, vp, vp, vp, i32, i32, vp, sz, vp]),
        "ssak_ctc_workspace_bytes": (sz, [i32, i32, i32, i32]),
        "ssak_ctc_loss_fwd_bwd": (i32, [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, f32, vp, vp, vp, vp, sz, vp]),
        "ssak_ctc_greedy_decode": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp]),
        "ssak_read_ranges": (i32, [vp, vp, vp, vp, i32, i32]),
        "ssak_drop_file_cache": (i32, [vp, i32]),
        "ssak_pcm_to_mono_f32": (i32, [vp, vp, vp, i32, i32, i32, i32, vp, vp]),
        "ssak_resample_plan": (i32, [i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]),
        "ssak_resample_table": (i32, [i32, i32, vp]),
        "ssak_resample_sinc": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, i32, vp, vp]),
        "ssak_ctc_wer_workspace_bytes": (sz, [i32, i32, i32]),
        "ssak_ctc_wer": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, sz, vp]),
        "ssak_ctc_align_workspace_bytes": (sz, [i32, i32]),
        "ssak_ctc_forced_align": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
        "ssak_ctc_align_batch_workspace_bytes": (sz, [i32, i32, i32]),
        "ssak_ctc_forced_align_batch": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp]),
        "ssak_gemm_bf16": (i32, [C.POINTER(GemmDesc), vp, vp, vp, vp, vp, vp, vp, sz, vp]),
        "ssak_gemm_f32": (i32, [C.POINTER(GemmDesc), vp, vp, vp, vp, vp, vp, vp]),
        "ssak_gemm_bf16_grouped": (i32, [C.POINTER(GemmDesc), i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), vp]),
        "ssak_gemm_fragment_b_bytes": (sz, [i32, i32]),
        "ssak_gemm_fragment_b": (i32, [vp, C.c_long, i32, i32, i32, vp, vp]),
        "ssak_gemm_fragment_b_batched": (i32, [i32, C.POINTER(vp), C.POINTER(C.c_long), C.POINTER(i32), C.POINTER(i32),
                                              C.POINTER(i32), C.POINTER(vp), vp]),
        "ssak_gemm_uses_fragments": (i32, [C.POINTER(GemmDesc)]),
        "ssak_conv0_workspace_bytes": (sz, [i32, i32, i32]),
        "ssak_conv0_gn_gelu": (i32, [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, vp]),
        "ssak_conv0_gn_gelu_raw": (i32, [vp, vp, vp, vp, vp, vp, sz, i32, i32, i32, vp]),
        "ssak_prof_enable": (i32, [vp, i32]),
        "ssak_prof_enable_slots": (i32, [vp, C.POINTER(C.c_int32), i32]),
        "ssak_prof_collect": (i32, [vp, C.POINTER(ProfEntry), i32]),
        "ssak_attention_fwd": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint64, C.c_uint32, vp]),
        "ssak_attention_bwd": (i32, [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint64, C.c_uint32, i32, vp]),
        "ssak_attention_bwd_bias_workspace_bytes": (sz, [i32, i32, i32]),
        "ssak_attention_bwd_bias": (i32, [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, C.c_uint64, C.c_uint32, vp, sz, vp]),
        "ssak_grad_sumsq": (i32, [vp, C.c_long, vp, vp, sz, vp]),
        "ssak_adamw_step": (i32, [vp, vp, vp, vp, vp, C.c_long, vp, f32, f32, f32, f32, f32, f32, f32, i32, vp]),
        "ssak_w2v2_create": (i32, [C.POINTER(W2V2Config), C.POINTER(vp)]),
        "ssak_w2v2_destroy": (None, [vp]),
        "ssak_w2v2_num_params": (C.c_long, [vp]),
        "ssak_w2v2_num_trainable": (C.c_long, [vp]),
        "ssak_w2v2_param_count": (i32, [vp]),
        "ssak_w2v2_param_info": (i32, [vp, i32, C.c_char_p, i32, C.POINTER(C.c_long), C.POINTER(C.c_long),
                                       C.POINTER(i32), C.POINTER(C.c_long)]),
        "ssak_w2v2_bind": (i32, [vp, vp, vp, vp]),
        "ssak_w2v2_sync_weights": (i32, [vp, i32, vp]),
        "ssak_w2v2_num_frames": (i32, [vp, i32]),
        "ssak_w2v2_workspace_bytes": (sz, [vp, i32, i32, i32]),
        "ssak_w2v2_forward": (i32, [vp, vp, vp, i32, i32, vp, vp, C.c_uint64, i32, vp, vp, vp, sz, vp]),
        "ssak_w2v2_backward": (i32, [vp, vp, vp, sz, vp]),
        "ssak_w2v2_set_grad_ready_callback": (i32, [vp, GRAD_READY_FN, vp]),
        "ssak_w2v2_set_param_event": (i32, [vp, vp, vp, vp]),
        "ssak_w2v2_set_option": (i32, [vp, i32, i32]),
        "ssak_w2v2_grad_ranges": (i32, [C.POINTER(W2V2Config), C.POINTER(C.c_long), C.POINTER(C.c_long), i32]),
        "ssak_w2v2_forward_hidden": (i32, [vp, vp, vp, i32, i32, vp, vp, C.c_uint64, i32, vp, vp, vp, sz, vp]),
        "ssak_w2v2_backward_hidden": (i32, [vp, vp, vp, sz, vp]),
        "ssak_grad_sumsq_add": (i32, [vp, C.c_long, vp, vp, sz, vp]),
        "ssak_utt_norm_workspace_bytes": (sz, [i32]),
        "ssak_utt_norm_fwd": (i32, [vp, vp, i32, C.c_long, i32, f32, vp, vp, sz, vp]),
        "ssak_utt_norm_bwd": (i32, [vp, vp, vp, i32, C.c_long, i32, vp, vp, sz, vp]),
        "ssak_batchnorm_workspace_bytes": (sz, [i32]),
        "ssak_batchnorm_act_fwd": (i32, [vp, vp, i32, i32, vp, vp, vp, vp, f32, f32, i32, f32, f32, C.c_uint64, C.c_uint32, vp, vp,
                                         vp, vp, sz, vp]),
        "ssak_batchnorm_act_bwd": (i32, [vp, vp, vp, i32, i32, vp, vp, vp, vp, f32, f32, C.c_uint64, C.c_uint32, vp, vp, vp, vp,
                                         vp, sz, vp]),
        "ssak_batchnorm_stats": (i32, [vp, i32, i32, vp, vp, sz, vp]),
        "ssak_adadelta_step": (i32, [vp, vp, vp, vp, vp, C.c_long, vp, f32, f32, f32, f32, f32, f32, vp]),
        "ssak_cast_f32_bf16": (i32, [vp, vp, C.c_long, vp]),
        "ssak_cast_bf16_f32": (i32, [vp, vp, C.c_long, vp]),
        "ssak_colsum_workspace_bytes": (sz, [i32]),
        "ssak_colsum_bf16": (i32, [vp, C.c_long, i32, i32, vp, vp, sz, vp]),
        "ssak_comm_unique_id": (i32, [vp]),
        "ssak_comm_create": (i32, [C.POINTER(vp), i32, i32, vp]),
        "ssak_allreduce": (i32, [vp, vp, C.c_long, C.c_long, i32, vp]),
        "ssak_comm_destroy": (i32, [vp]),
        "ssak_debug_dropout_mask": (i32, [C.c_uint64, C.c_uint32, f32, C.c_long, i32, vp, C.POINTER(f32), vp]),
        "ssak_debug_attention_dropout_mask": (i32, [C.c_uint64, C.c_uint32, f32, i32, i32, i32, vp, vp]),
    }
    lib.ssak_version.restype = i32
    got = lib.ssak_version()
    if got != ABI_VERSION:
        # struct layouts (GemmDesc, W2V2Config, ProfEntry) and signatures belong to ONE ABI: driving another build through
        # this table would pass arguments at the wrong offsets, so refuse it instead of skipping what it lacks
        raise ImportError(f"{_LIB_PATH} reports ABI {got}, this binding is written for {ABI_VERSION}: rebuild with `make`"
                          + (" (SSAK_HIP_LIB must name a build of the same ABI)" if os.environ.get("SSAK_HIP_LIB") else ""))
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


lib = _load()
SSAK_OK, SSAK_ERR_INVALID, SSAK_ERR_LAUNCH, SSAK_ERR_STATE = 0, -1, -2, -3
EPI_NONE, EPI_GELU, EPI_MUL_GELU_GRAD, EPI_GELU_SAVE_GRAD, EPI_MUL_AUX = 0, 1, 2, 3, 4
REDUCTION = {"sum": 0, "mean": 1}


def check(rc: int):
    if rc == SSAK_OK:
        return
    msg = lib.ssak_last_error().decode()
    if rc == SSAK_ERR_INVALID:
        raise ValueError(msg)
    if rc == SSAK_ERR_STATE:
        raise RuntimeError(msg)
    raise RuntimeError(f"libssak_hip status {rc}: {msg}")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ws(nbytes: int, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------ op wrappers (tensors in, tensors out)
def wave_normalize(x: torch.Tensor, lens: torch.Tensor | None = None, return_mask: bool = False):
    """[B,T] fp32 raw samples -> normalised [B,T] fp32 (+ int32 attention mask)."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
    B, T = x.shape
    out = torch.empty_like(x)
    mask = torch.empty((B, T), dtype=torch.int32, device=x.device) if return_mask else None
    if lens is not None:
        lens = lens.to(device=x.device, dtype=torch.int32).contiguous()
    ws = _ws(lib.ssak_wave_normalize_workspace_bytes(B, T), x.device)
    check(lib.ssak_wave_normalize(ptr(x), ptr(lens), B, T, ptr(out), ptr(mask), ptr(ws), ws.numel(), stream()))
    return (out, mask) if return_mask else out


_LOGMEL_TABLES = {}


def logmel_tables(device):
    key = str(device)
    if key not in _LOGMEL_TABLES:
        t = torch.empty(lib.ssak_logmel_table_floats(), dtype=torch.float32, device=device)
        check(lib.ssak_logmel_init_tables(ptr(t)))
        _LOGMEL_TABLES[key] = t
    return _LOGMEL_TABLES[key]


def logmel_whisper(wav: torch.Tensor, lens: torch.Tensor | None = None, n_samples: int = 480000,
                   channels_last: torch.Tensor | None = None, cl_lead: int = 0, want_mel: bool = True):
    """[B,T] fp32 waveforms -> Whisper input features [B, 80, n_samples/160] fp32 (30 s windows by default)."""
    assert wav.is_cuda and wav.dtype == torch.float32 and wav.dim() == 2 and wav.is_contiguous()
    B, T = wav.shape
    if lens is not None:
        lens = lens.to(device=wav.device, dtype=torch.int32).contiguous()
    mel = torch.empty((B, 80, n_samples // 160), dtype=torch.float32, device=wav.device) if want_mel else None
    ws = _ws(lib.ssak_logmel_workspace_bytes(B, n_samples), wav.device)
    cl_rows = 0 if channels_last is None else channels_last.shape[1]
    check(lib.ssak_logmel_whisper(ptr(wav), ptr(lens), B, T, n_samples, ptr(logmel_tables(wav.device)), ptr(mel),
                                  ptr(channels_last), cl_rows, cl_lead, ptr(ws), ws.numel(), stream()))
    return mel


def ctc_loss(logits: torch.Tensor, in_lens: torch.Tensor | None, labels: torch.Tensor, blank: int = 0,
             reduction: str = "mean", zero_infinity: bool = True, grad_scale: float = 1.0, want_grad: bool = True):
    """logits [B,F,V] fp32, labels [B,L] (negative = pad) -> (loss[1], nll[B], dlogits[B,F,V] | None)."""
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 3 and logits.is_contiguous()
    B, F, V = logits.shape
    if not labels.is_cuda and labels.numel() and int(labels.max()) >= V:
        raise ValueError(f"Label values must be <= vocab_size: {V}")  # modeling_wav2vec2.py:1686-1687
    # device-resident labels are validated by the kernel instead (bad label -> NaN nll): no host sync per step
    labels = labels.to(device=logits.device, dtype=torch.int32).contiguous()
    Lmax = labels.shape[1]
    if in_lens is not None:
        in_lens = in_lens.to(device=logits.device, dtype=torch.int32).contiguous()
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    nll = torch.empty(B, dtype=torch.float32, device=logits.device)
    grad = torch.empty_like(logits) if want_grad else None
    ws = _ws(lib.ssak_ctc_workspace_bytes(B, F, V, Lmax), logits.device)
    check(lib.ssak_ctc_loss_fwd_bwd(ptr(logits), ptr(in_lens), ptr(labels), B, F, V, Lmax, blank, REDUCTION[reduction],
                                    int(zero_infinity), float(grad_scale), ptr(loss), ptr(nll), ptr(grad), ptr(ws),
                                    ws.numel(), stream()))
    return loss, nll, grad


def ctc_greedy_decode(logits: torch.Tensor, in_lens: torch.Tensor | None = None, blank: int = 0):
    assert logits.is_cuda and logits.dtype == torch.float32 and logits.dim() == 3 and logits.is_contiguous()
    B, F, V = logits.shape
    if in_lens is not None:
        in_lens = in_lens.to(device=logits.device, dtype=torch.int32).contiguous()
    ids = torch.empty((B, F), dtype=torch.int32, device=logits.device)
    n = torch.empty(B, dtype=torch.int32, device=logits.device)
    check(lib.ssak_ctc_greedy_decode(ptr(logits), ptr(in_lens), B, F, V, blank, ptr(ids), ptr(n), stream()))
    return ids, n


def ctc_wer(hyp_ids: torch.Tensor, hyp_lens: torch.Tensor, labels: torch.Tensor, token_class: torch.Tensor):
    """(edits [B], ref_words [B]) int32 on the device; see ``ssak_ctc_wer`` in include/ssak_hip.h."""
    B, F = hyp_ids.shape
    labels = labels.to(device=hyp_ids.device, dtype=torch.int32).contiguous()
    token_class = token_class.to(device=hyp_ids.device, dtype=torch.uint8).contiguous()
    Lmax, V = labels.shape[1], token_class.numel()
    edits = torch.empty(B, dtype=torch.int32, device=hyp_ids.device)
    nref = torch.empty(B, dtype=torch.int32, device=hyp_ids.device)
    ws = _ws(lib.ssak_ctc_wer_workspace_bytes(B, F, Lmax), hyp_ids.device)
    check(lib.ssak_ctc_wer(ptr(hyp_ids), ptr(hyp_lens), ptr(labels), ptr(token_class), B, F, Lmax, V, ptr(edits), ptr(nref),
                           ptr(ws), ws.numel(), stream()))
    return edits, nref


def gemm(A, B, C_out, M, N, K, *, a_kmajor=False, b_kmajor=False, lda=None, ldb=None, ldc=None, nb1=1, nb2=1,
         sa=(0, 0), sb=(0, 0), sc=(0, 0), alpha=1.0, bias=None, epilogue=EPI_NONE, aux_in=None, aux_out=None,
         accumulate=False, split_k=1, drop_p=0.0, drop_stream=0, drop_seed=0, pads_are_zero=False, colsum_out=None,
         dynamic_tiles=False, b_fragments=None, plan_tile=0):
    """Raw descriptor-level GEMM on device tensors (see ``ssak_gemm_desc`` in include/ssak_hip.h).  ``b_fragments``: the
    copy of B made by :func:`gemm_fragment_b` (used when the library picks the B-direct kernel, ignored otherwise)."""
    d = GemmDesc(M, N, K, int(a_kmajor), int(b_kmajor), lda, ldb, ldc, nb1, nb2, sa[0], sa[1], sb[0], sb[1], sc[0],
                 sc[1], float(alpha), epilogue, int(C_out.dtype == torch.float32), int(accumulate), split_k, float(drop_p),
                 drop_stream, drop_seed, 0, int(pads_are_zero), int(colsum_out is not None), int(dynamic_tiles),
                 ptr(b_fragments), int(plan_tile))
    n_slabs = split_k if split_k > 0 else max(1, min(32, ((K + 63) // 64) // 4))  # 0 = library-sized split
    ws = _ws(n_slabs * nb1 * nb2 * M * N * 4, A.device) if n_slabs > 1 else None
    if colsum_out is not None:
        aux_out = colsum_out
        ws = _ws(((M + 63) // 64) * N * 4, A.device)
    check(lib.ssak_gemm_bf16(C.byref(d), ptr(A), ptr(B), ptr(C_out), ptr(bias), ptr(aux_in), ptr(aux_out), ptr(ws),
                             0 if ws is None else ws.numel(), stream()))
    return C_out


def gemm_fragment_b(B, N, K, *, ldb=None, b_kmajor=False):
    """Fragment-ordered copy of a weight operand (``ssak_gemm_fragment_b``): a bf16 tensor of ``ssak_gemm_fragment_b_bytes``."""
    ldb = ldb if ldb is not None else (N if b_kmajor else K)
    out = torch.empty(lib.ssak_gemm_fragment_b_bytes(N, K) // 2, dtype=torch.bfloat16, device=B.device)
    check(lib.ssak_gemm_fragment_b(ptr(B), ldb, N, K, int(b_kmajor), ptr(out), stream()))
    return out


def gemm_uses_fragments(M, N, K, *, a_kmajor=False, b_kmajor=False, lda=None, ldb=None, ldc=None, pads_are_zero=False):
    d = GemmDesc(M, N, K, int(a_kmajor), int(b_kmajor), lda if lda is not None else (M if a_kmajor else K),
                 ldb if ldb is not None else (N if b_kmajor else K), ldc if ldc is not None else N, 1, 1, 0, 0, 0, 0, 0, 0, 1.0,
                 EPI_NONE, 0, 0, 1, 0.0, 0, 0, 0, int(pads_are_zero), 0, 0, None)
    return bool(lib.ssak_gemm_uses_fragments(C.byref(d)))


def gemm_f32(A, B, C_out, M, N, K, *, a_kmajor=False, b_kmajor=False, lda=None, ldb=None, ldc=None, nb1=1, nb2=1,
             sa=(0, 0), sb=(0, 0), sc=(0, 0), alpha=1.0, bias=None, epilogue=EPI_NONE, aux_in=None, aux_out=None,
             accumulate=False, drop_p=0.0, drop_stream=0, drop_seed=0, colsum_out=None, bias_s2=0):
    """The fp32 GEMM of the exact mode (``ssak_gemm_f32``): same descriptor as :func:`gemm`, float tensors."""
    d = GemmDesc(M, N, K, int(a_kmajor), int(b_kmajor), lda, ldb, ldc, nb1, nb2, sa[0], sa[1], sb[0], sb[1], sc[0],
                 sc[1], float(alpha), epilogue, 1, int(accumulate), 1, float(drop_p), drop_stream, drop_seed, bias_s2, 0,
                 int(colsum_out is not None), 0)
    if colsum_out is not None:
        aux_out = colsum_out
    check(lib.ssak_gemm_f32(C.byref(d), ptr(A), ptr(B), ptr(C_out), ptr(bias), ptr(aux_in), ptr(aux_out), stream()))
    return C_out


def gemm_grouped(problems, stream_=None, dynamic_tiles=False):
    """problems: list of (A, B, C_out, M, N, K, lda, ldb, ldc) sharing K / layouts (a_kmajor, b_kmajor passed per call via
    keyword in each tuple's dict is not needed: the weight-gradient form is k-major on both operands)."""
    n = len(problems)
    descs = (GemmDesc * n)()
    pa, pb, pc = (C.c_void_p * n)(), (C.c_void_p * n)(), (C.c_void_p * n)()
    for i, (A, B, Cout, M, N, K, lda, ldb, ldc, akm, bkm) in enumerate(problems):
        descs[i] = GemmDesc(M, N, K, int(akm), int(bkm), lda, ldb, ldc, 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0,
                            int(Cout.dtype == torch.float32), 0, 1, 0.0, 0, 0, 0, 1, 0, int(dynamic_tiles))
        pa[i], pb[i], pc[i] = A.data_ptr(), B.data_ptr(), Cout.data_ptr()
    check(lib.ssak_gemm_bf16_grouped(descs, n, pa, pb, pc, stream()))


DTYPE_F32, DTYPE_BF16 = 0, 1


class Comm:
    """RCCL communicator behind the C ABI (``ssak_comm_*`` / ``ssak_allreduce``): what a host without torch.distributed uses for
    the data-parallel exchange.  ``unique_id()`` on rank 0, the 128 bytes carried to the other ranks by the host's own means,
    then ``Comm(world, rank, id)`` on every rank (collective)."""

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(128)
        check(lib.ssak_comm_unique_id(buf))
        return buf.raw

    def __init__(self, world: int, rank: int, uid: bytes):
        assert len(uid) == 128
        self._h = C.c_void_p()
        check(lib.ssak_comm_create(C.byref(self._h), int(world), int(rank), C.create_string_buffer(uid, 128)))
        self.world, self.rank = world, rank

    def all_reduce(self, t: torch.Tensor, offset: int = 0, count: int | None = None, stream_=None):
        """In-place sum over ranks of ``t.view(-1)[offset : offset + count]`` (fp32 or bf16), asynchronous on the current stream."""
        dt = {torch.float32: DTYPE_F32, torch.bfloat16: DTYPE_BF16}[t.dtype]
        n = t.numel() - offset if count is None else count
        check(lib.ssak_allreduce(self._h, ptr(t), int(offset), int(n), dt, stream() if stream_ is None else stream_))

    def close(self):
        if self._h:
            check(lib.ssak_comm_destroy(self._h))
            self._h = C.c_void_p()


def prof_enable(mode: int):
    """Launch timing on the CURRENT stream: 0 = off, 1 = every launch, 2 + i = only the slot at index i of
    :func:`prof_collect`'s list.  Per stream: other streams / handles are neither slowed nor recorded."""
    check(lib.ssak_prof_enable(stream(), int(mode)))


def prof_enable_slots(slots):
    """Launch timing on the CURRENT stream for the slots at these indices of :func:`prof_collect`'s list only."""
    arr = (C.c_int32 * len(slots))(*[int(v) for v in slots])
    check(lib.ssak_prof_enable_slots(stream(), arr, len(slots)))


BOUNDS = {0: "mfma", 1: "hbm", 2: "latency"}


def prof_collect():
    """[(kernel name, launches, total ms, total algorithmic work, bound)] since the last collect; work = flops for "mfma"
    slots, bytes for "hbm" / "latency" slots."""
    arr = (ProfEntry * 128)()
    n = lib.ssak_prof_collect(stream(), arr, 128)
    if n < 0:
        check(n)
    return [(arr[i].name.decode(), arr[i].launches, arr[i].total_ms, arr[i].total_flops, BOUNDS[arr[i].bound]) for i in range(n)]


def conv0_gn_gelu(x: torch.Tensor, w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, raw: bool = False):
    """x [B, T] fp32, w [C, 10], gamma / beta [C] -> [B, T0, C] bf16 (conv k=10 s=5 -> GroupNorm per channel -> GELU).  ``raw``:
    x are RAW full-length waveforms, the zero-mean / unit-variance normalisation folded into the GroupNorm statistics."""
    B, T = x.shape
    C = w.shape[0]
    T0 = (T - 10) // 5 + 1
    out = torch.full((B, T0, C), float("nan"), dtype=torch.bfloat16, device=x.device)
    nb = lib.ssak_conv0_workspace_bytes(B, T, C)
    ws = torch.empty(nb, dtype=torch.uint8, device=x.device)
    fn = lib.ssak_conv0_gn_gelu_raw if raw else lib.ssak_conv0_gn_gelu
    check(fn(ptr(x), ptr(w.contiguous()), ptr(gamma), ptr(beta), ptr(out), ptr(ws), nb, B, T, C, stream()))
    return out


def attention_fwd(qkv: torch.Tensor, B: int, F: int, nh: int, klens=None, drop_p=0.0, seed=0, stream_id=0):
    """qkv [B*F, 3H] bf16 -> (ctx [B*F, H] bf16, lse [B, nh, F] fp32)."""
    H = qkv.shape[1] // 3
    ctx = torch.empty((B * F, H), dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty((B, nh, F), dtype=torch.float32, device=qkv.device)
    if klens is not None:
        klens = klens.to(device=qkv.device, dtype=torch.int32).contiguous()
    check(lib.ssak_attention_fwd(ptr(qkv), ptr(ctx), ptr(lse), ptr(klens), B, F, nh, H, float(drop_p), seed, stream_id, stream()))
    return ctx, lse


ATTN_BWD_DEFAULT, ATTN_BWD_TWO_KERNEL = 0, 1
W2V2_OPT_DYNAMIC_TILES, W2V2_OPT_ATTENTION_BWD, W2V2_OPT_POSCONV_DIRECT, W2V2_OPT_FRAGMENT_WEIGHTS, W2V2_OPT_TRANSPOSED_WEIGHTS = 1, 2, 3, 4, 5
W2V2_OPT_RAW_INPUT = 6


def attention_bwd(qkv, ctx, lse, dctx, B: int, F: int, nh: int, klens=None, drop_p=0.0, seed=0, stream_id=0,
                  mode=ATTN_BWD_DEFAULT):
    """``mode``: ATTN_BWD_DEFAULT = ATTN_BWD_TWO_KERNEL (dQ; dK + dV); the single-pass form (2) was removed in ABI 400."""
    H = qkv.shape[1] // 3
    dqkv = torch.full_like(qkv, float("nan"))  # poisoned: the kernels write every element
    delta = torch.empty((B, nh, F), dtype=torch.float32, device=qkv.device)
    if klens is not None:
        klens = klens.to(device=qkv.device, dtype=torch.int32).contiguous()
    check(lib.ssak_attention_bwd(ptr(qkv), ptr(ctx), ptr(lse), ptr(klens), ptr(dctx), ptr(delta), ptr(dqkv), B, F, nh, H,
                                 float(drop_p), seed, stream_id, int(mode), stream()))
    return dqkv


def attention_bwd_bias(qkv, ctx, lse, dctx, B: int, F: int, nh: int, klens=None, drop_p=0.0, seed=0, stream_id=0, bias_grad=None):
    """attention_bwd + the q|k|v projection bias gradient (column sums of dqkv, taken inside the kernels): ``bias_grad`` [3H] fp32
    is ADDED to (zeros when None).  Returns (dqkv, bias_grad)."""
    H = qkv.shape[1] // 3
    dqkv = torch.full_like(qkv, float("nan"))
    delta = torch.empty((B, nh, F), dtype=torch.float32, device=qkv.device)
    if bias_grad is None:
        bias_grad = torch.zeros(3 * H, dtype=torch.float32, device=qkv.device)
    if klens is not None:
        klens = klens.to(device=qkv.device, dtype=torch.int32).contiguous()
    nbytes = lib.ssak_attention_bwd_bias_workspace_bytes(B, F, H)
    ws = _ws(nbytes, qkv.device)
    check(lib.ssak_attention_bwd_bias(ptr(qkv), ptr(ctx), ptr(lse), ptr(klens), ptr(dctx), ptr(delta), ptr(dqkv), ptr(bias_grad), B, F,
                                      nh, H, float(drop_p), seed, stream_id, ptr(ws), nbytes, stream()))
    return dqkv, bias_grad
