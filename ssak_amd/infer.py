"""Inference entry point: the counterpart of ``ssak/infer/transformers_infer.py`` (console script ``sak_infer``)
with the acoustic model on HIP kernels.  Same functions, arguments and CLI flags
(transformers_infer.py:14-133,190-269,316-365); ``--arpa`` (KenLM beam search on the CPU) is outside this path.
"""
from __future__ import annotations

import os
import sys
from typing import List, Optional

import numpy as np
import torch

from . import hip
from .checkpoint import load_pretrained
from .data import to_audio_batches

MAX_SAMPLES = 2240400  # chunking threshold of the reference (transformers_infer.py:190)


def transformers_load_model(source, device=None):
    """Model folder (HF layout) -> (model, tokenizer); a (model, tokenizer) pair passes through."""
    if isinstance(source, (tuple, list)):
        return source
    device = device or "cuda:0"
    model, tok = load_pretrained(source, device=device)
    return model.eval(), tok


def transformers_compute_logits(model, processor, batch: List[np.ndarray], device=None, language=None,
                                sample_rate: int = 16000, max_duration: int = MAX_SAMPLES) -> torch.Tensor:
    """list of float32 waveforms -> CPU fp32 logits [B, F, V] (transformers_infer.py:190-269): normalise + pad
    (a1, on the device), forward; inputs longer than ``max_duration`` samples are split on the sample axis and the
    logits concatenated on the frame axis (:259-265).  Group-norm ("base") models run without attention mask, as HF
    prescribes for them; layer-norm models get the lengths."""
    lens = np.array([len(a) for a in batch], dtype=np.int32)
    T = int(lens.max())
    x = np.zeros((len(batch), T), dtype=np.float32)
    for i, a in enumerate(batch):
        x[i, :len(a)] = a
    dev = model.device
    xd = torch.from_numpy(x).to(dev)
    ld = torch.from_numpy(lens).to(dev)
    with torch.cuda.device(dev):
        xn = hip.wave_normalize(xd, ld)
    use_mask = model.config.feat_extract_norm == "layer"
    outs = []
    for s in range(0, T, max_duration):
        chunk = xn[:, s:s + max_duration].contiguous()
        if model.num_frames(chunk.shape[1]) <= 0:
            continue
        cl = (ld - s).clamp(min=0, max=chunk.shape[1]) if use_mask else None
        outs.append(model(chunk, lengths=cl).logits)
    return torch.cat(outs, dim=1).cpu()


def compute_logits(model_and_processor, audio: np.ndarray, sample_rate: int = 16000) -> torch.Tensor:
    """Single utterance -> [F, V] (the dispatcher contract of ssak/infer/general.py:76-97)."""
    model, proc = model_and_processor
    return transformers_compute_logits(model, proc, [audio], sample_rate=sample_rate)[0]


def compute_log_probas(model_and_processor, audio: np.ndarray, sample_rate: int = 16000) -> torch.Tensor:
    """log_softmax of the logits (ssak/infer/general.py:99-101)."""
    return torch.log_softmax(compute_logits(model_and_processor, audio, sample_rate), dim=-1)


def transformers_infer(source, audios, batch_size: int = 1, device=None, language=None, arpa_path=None,
                       alpha: float = 0.5, beta: float = 1.0, sort_by_len: bool = False, output_ids: bool = False,
                       log_memtime: bool = False):
    """Generator of transcripts (or (id, transcript)) -- greedy CTC path of transformers_infer.py:73-95."""
    if arpa_path is not None:
        raise NotImplementedError("--arpa: n-gram LM beam search (pyctcdecode/kenlm, CPU) is outside the HIP path")
    model, tok = transformers_load_model(source, device)
    for batch in to_audio_batches(audios, batch_size=batch_size, sort_by_len=sort_by_len, output_ids=output_ids):
        ids = None
        if output_ids:
            ids = [b[1] for b in batch]
            batch = [b[0] for b in batch]
        logits = transformers_compute_logits(model, tok, batch).to(model.device).contiguous()
        # argmax + collapse on the device (a12); only the short id strings cross PCIe
        lens = torch.tensor([model.num_frames(len(a)) for a in batch], dtype=torch.int32)
        with torch.cuda.device(model.device):
            dec, n = hip.ctc_greedy_decode(logits, lens, tok.pad_token_id)
        dec, n = dec.cpu().numpy(), n.cpu().numpy()
        for i in range(len(batch)):
            text = tok.decode(dec[i, :n[i]], group_tokens=False)
            yield (ids[i], text) if output_ids else text


def cli(argv: Optional[List[str]] = None):
    import argparse
    p = argparse.ArgumentParser(description="Transcribe audio(s) with a wav2vec2 CTC model on MI355X (HIP kernels)",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("data", help="Path to data (audio file(s) or kaldi folder(s))", nargs="+")
    p.add_argument("--model", help="Path to trained folder", required=True)
    p.add_argument("--language", default=None, type=str)
    p.add_argument("--arpa", help="Path to a n-gram language model", default=None)
    p.add_argument("--output", help="Output path (will print on stdout by default)", default=None)
    p.add_argument("--use_ids", default=False, action="store_true")
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--gpus", default=None)
    p.add_argument("--sort_by_len", default=False, action="store_true")
    p.add_argument("--enable_logs", default=False, action="store_true")
    args = p.parse_args(argv)
    if args.gpus:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", str(args.gpus))
    out = sys.stdout
    if args.output == "/dev/null":
        out = open(os.devnull, "w")
    elif args.output:
        d = os.path.dirname(args.output)
        if d and not os.path.isdir(d):
            os.makedirs(d)
        out = open(args.output, "w")
    for reco in transformers_infer(args.model, args.data, batch_size=args.batch_size, sort_by_len=args.sort_by_len,
                                   output_ids=args.use_ids, language=args.language, arpa_path=args.arpa,
                                   log_memtime=args.enable_logs):
        print(reco if isinstance(reco, str) else " ".join(reco), file=out)
        out.flush()


if __name__ == "__main__":
    cli()
