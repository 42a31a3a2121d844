"""The SpeechBrain recipe's command line on the device: counterpart of ``ssak/train/speechbrain/wav2vec_train.py:496-650``.

    python -m ssak_amd.train_speechbrain HPARAMS.yaml --train=KALDI --valid=KALDI --base_model=FOLDER [--key=value ...]

``HPARAMS.yaml`` is the recipe's own file (``ssak/train/speechbrain/fr/hyperparameters_wav2vec_finetune_cv-fr.yaml``): its
scalar entries (num_epochs, lr, lr_wav2vec, batch_size, test_batch_size, min/max_duration, freeze_wav2vec, eval_steps, seed,
sorting, dnn_neurons, output_neurons, blank_index ...) and the NewBob / Adadelta blocks are read with a tag-tolerant YAML
loader (hyperpyyaml itself is not installed: ``!new:`` / ``!name:`` nodes are taken as plain mappings, ``!ref <key>`` is
resolved for scalars, ``!PLACEHOLDER`` must come from the command line) and every ``--key=value`` overrides an entry, as
``sb.parse_arguments`` does.  What the objects of the yaml stand for is built by ``ssak_amd.sb_head`` (Brain, CTCHead,
Adadelta, NewBobScheduler).  Differences, all on the host side of the path: ``base_model`` is a HuggingFace-layout wav2vec2
folder (speechbrain checkpoints cannot be read without speechbrain), the tokenizer is the character set of the training
text with index 0 = blank (the recipe trains a SentencePiece "char" model of ``output_neurons`` pieces on the same text),
``TimeDomainSpecAugment`` is not applied.

Output folder (``<output_folder_prefix>sb_<md5 of data>_...``): ``train_log.txt`` (one line per validation, the recipe's
fields), ``save/CKPT-<step>/`` (head, optimizers, schedulers, wav2vec2 when unfrozen; the two best by WER are kept, as
``save_and_keep_only(min_keys=["WER"])`` keeps the best), ``final/`` = the best checkpoint; a rerun resumes from the newest
checkpoint (:625-640).
"""
from __future__ import annotations

import json
import os
import re
import shutil
import sys
import time
from typing import Dict, List

import numpy as np
import torch
import yaml

from .data import load_audio, load_kaldi, pad_waves, remove_special_words
from .naming import hashmd5


# ------------------------------------------------------------------ yaml with hyperpyyaml tags
class _Loader(yaml.SafeLoader):
    pass


def _tagged(loader, suffix, node):
    if isinstance(node, yaml.MappingNode):
        return loader.construct_mapping(node, deep=True)
    if isinstance(node, yaml.SequenceNode):
        return loader.construct_sequence(node, deep=True)
    v = loader.construct_scalar(node)
    if suffix.startswith("ref"):
        return "!ref " + v
    if suffix.startswith("PLACEHOLDER"):
        return "!PLACEHOLDER"
    return v


_Loader.add_multi_constructor("!", _tagged)


def load_hparams(path: str, overrides: Dict[str, str]) -> dict:
    with open(path) as f:
        hp = yaml.load(f, Loader=_Loader) or {}
    for k, v in overrides.items():
        old = hp.get(k)
        if isinstance(old, bool):
            hp[k] = str(v).lower() in ("1", "true", "yes")
        elif isinstance(old, int) and not isinstance(old, bool):
            hp[k] = int(float(v))
        elif isinstance(old, float):
            hp[k] = float(v)
        else:
            hp[k] = yaml.safe_load(v) if not isinstance(v, str) or re.fullmatch(r"[-+0-9.eE]+|true|false|True|False", v) else v

    def resolve(v, depth=0):
        if isinstance(v, str) and v.startswith("!ref ") and depth < 8:
            expr = v[5:]
            m = re.fullmatch(r"<([A-Za-z0-9_]+)>", expr.strip())
            if m:  # a plain reference keeps the referenced type
                return resolve(hp.get(m.group(1)), depth + 1)
            return re.sub(r"<([A-Za-z0-9_]+)>", lambda mm: str(resolve(hp.get(mm.group(1)), depth + 1)), expr)
        if isinstance(v, dict):
            return {k: resolve(x, depth) for k, x in v.items()}
        return v

    hp = {k: resolve(v) for k, v in hp.items()}
    missing = [k for k, v in hp.items() if v == "!PLACEHOLDER"]
    if missing:
        raise SystemExit(f"mandatory entries without a value: {', '.join('--' + k for k in missing)}")
    return hp


def parse_argv(argv: List[str]):
    """(yaml file, {key: value}) from ``FILE --key=value --key value --flag`` (sb.parse_arguments' override syntax)."""
    files, overrides, i = [], {}, 0
    while i < len(argv):
        a = argv[i]
        if not a.startswith("--"):
            files.append(a)
        elif "=" in a:
            k, v = a[2:].split("=", 1)
            overrides[k] = v
        elif i + 1 < len(argv) and not argv[i + 1].startswith("--"):
            overrides[a[2:]] = argv[i + 1]
            i += 1
        else:
            overrides[a[2:]] = "true"
        i += 1
    overrides.pop("gpus", None)  # (--gpus is stripped by the reference too, :499-507)
    if len(files) != 1:
        raise SystemExit("usage: python -m ssak_amd.train_speechbrain HPARAMS.yaml --train=FOLDER --valid=FOLDER --base_model=FOLDER "
                         "[--key=value ...]")
    return files[0], overrides


# ------------------------------------------------------------------ data
class CharVocab:
    """Index 0 = blank, 1 = space, then the characters of the training text in sorted order."""

    def __init__(self, symbols: List[str]):
        self.symbols = symbols
        self.index = {s: i for i, s in enumerate(symbols)}

    @classmethod
    def from_texts(cls, texts):
        chars = sorted({c for t in texts for c in t if c != " "})
        return cls(["<blank>", " "] + chars)

    def encode(self, text: str) -> List[int]:
        return [self.index[c] for c in text if c in self.index]


def batches_of(order: List[int], bs: int):
    return [order[i:i + bs] for i in range(0, len(order), bs)]


def pad_tokens(tok_lists):
    L = max(1, max(len(t) for t in tok_lists))
    out = np.zeros((len(tok_lists), L), dtype=np.int64)
    for i, t in enumerate(tok_lists):
        out[i, :len(t)] = t
    return out, np.array([len(t) / L for t in tok_lists], dtype=np.float32)


# ------------------------------------------------------------------ checkpoints
def _ckpt_dirs(save_dir):
    if not os.path.isdir(save_dir):
        return []
    ds = [d for d in os.listdir(save_dir) if d.startswith("CKPT-") and os.path.exists(os.path.join(save_dir, d, "meta.json"))]
    return [os.path.join(save_dir, d) for d in sorted(ds, key=lambda d: int(d[5:]))]


def save_checkpoint(save_dir, step, brain, meta, keep_best=2):
    d = os.path.join(save_dir, f"CKPT-{step}")
    os.makedirs(d, exist_ok=True)
    torch.save(brain.head.state_dict(), os.path.join(d, "model.ckpt"))
    torch.save({"modelopt": brain.model_optimizer.state_dict(),
                "wav2vec_opt": None if brain.wav2vec_optimizer is None else brain.wav2vec_optimizer.state_dict(),
                "scheduler_model": brain.lr_annealing_model.state_dict(), "scheduler_wav2vec": brain.lr_annealing_wav2vec.state_dict(),
                "head_seed": brain.head._seed, "step_seed": int(brain.wav2vec2._step_seed)}, os.path.join(d, "optim.ckpt"))
    if not brain.freeze:
        torch.save(brain.wav2vec2.state_dict(), os.path.join(d, "wav2vec2.ckpt"))
    with open(os.path.join(d, "meta.json"), "w") as f:
        json.dump(meta, f)
    # keep the newest (to resume from) and the best by WER (save_and_keep_only(min_keys=["WER"]), :212-214)
    cks = _ckpt_dirs(save_dir)
    metas = {c: json.load(open(os.path.join(c, "meta.json"))) for c in cks}
    best = sorted(cks, key=lambda c: (metas[c]["WER"], -metas[c]["step"]))[:keep_best - 1]
    for c in cks:
        if c != cks[-1] and c not in best:
            shutil.rmtree(c, ignore_errors=True)
    return d


def load_checkpoint(d, brain):
    brain.head.load_state_dict(torch.load(os.path.join(d, "model.ckpt")))
    o = torch.load(os.path.join(d, "optim.ckpt"), weights_only=False)
    brain.model_optimizer.load_state_dict(o["modelopt"])
    brain.lr_annealing_model.load_state_dict(o["scheduler_model"])
    brain.lr_annealing_wav2vec.load_state_dict(o["scheduler_wav2vec"])
    brain.head._seed, brain.wav2vec2._step_seed = o["head_seed"], o["step_seed"]
    if brain.wav2vec_optimizer is not None and o["wav2vec_opt"] is not None:
        brain.wav2vec_optimizer.load_state_dict(o["wav2vec_opt"])
        brain.wav2vec_optimizer.lr = brain.lr_annealing_wav2vec.hyperparam_value
    if not brain.freeze and os.path.exists(os.path.join(d, "wav2vec2.ckpt")):
        brain.wav2vec2.load_state_dict(torch.load(os.path.join(d, "wav2vec2.ckpt")))
    return json.load(open(os.path.join(d, "meta.json")))


# ------------------------------------------------------------------ main
def main(argv=None):
    hfile, overrides = parse_argv(list(sys.argv[1:] if argv is None else argv))
    hp = load_hparams(hfile, overrides)
    if not hp.get("base_model") or not os.path.isdir(str(hp["base_model"])):
        raise SystemExit("--base_model must be a HuggingFace-layout wav2vec2 folder (config.json, model.safetensors, vocab.json)")
    from .checkpoint import load_pretrained
    from .sb_head import VALID, Brain, CTCHead
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    seed = int(hp.get("seed", 1234))
    rng = np.random.RandomState(seed)
    train_u = load_kaldi(hp["train"], hp.get("min_duration", 0.5), hp.get("max_duration", 15))
    valid_u = load_kaldi(hp["valid"], max(1, hp.get("min_duration", 0.5)) if not hp.get("debug") else 0, hp.get("max_duration", 15))[:480]
    if hp.get("debug"):
        nb = int(hp.get("debug_num_batches", 3))
        train_u, valid_u = train_u[:nb * hp["batch_size"]], valid_u[:2 * hp["test_batch_size"]]
    texts = [remove_special_words(u.text) for u in train_u]
    vocab = CharVocab.from_texts(texts)
    n_out = max(int(hp.get("output_neurons", 0)), len(vocab.symbols))
    name = (f"sb_{hashmd5([u.path for u in train_u])[:8]}_len-{hp.get('min_duration')}-{hp.get('max_duration')}_fr{hp.get('freeze_wav2vec')}"
            f"_lr{hp.get('lr')}-{hp.get('lr_wav2vec')}_bs{hp.get('batch_size')}_s{seed}_{hp.get('sorting', 'random')}")
    out_dir = str(hp.get("output_folder_prefix", "") or "") + name  # a prefix, not a folder (yaml :44-45)
    save_dir = os.path.join(out_dir, "save")
    freeze = bool(hp.get("freeze_wav2vec", True))
    w2v2, _tok = load_pretrained(hp["base_model"], device=dev, freeze_feature_encoder=True, mask_time_prob=0.0)
    head = CTCHead(w2v2.config.hidden_size, int(hp.get("dnn_neurons", 1024)), n_out, device=dev, seed=seed)
    ann = (hp.get("lr_annealing_model") or {}, hp.get("lr_annealing_wav2vec") or {})
    brain = Brain(w2v2, head, freeze_wav2vec=freeze, lr=float(hp.get("lr", 1.0)), lr_wav2vec=float(hp.get("lr_wav2vec", 1e-4)),
                  blank_index=int(hp.get("blank_index", 0)), vocab=vocab.symbols + ["<unused>"] * (n_out - len(vocab.symbols)),
                  annealing=(float(ann[0].get("annealing_factor", 0.8)), float(ann[1].get("annealing_factor", 0.9))),
                  improvement_threshold=float(ann[0].get("improvement_threshold", 0.0025)))
    opt_cfg = hp.get("model_opt_class") or {}
    brain.model_optimizer.rho, brain.model_optimizer.eps = float(opt_cfg.get("rho", 0.95)), float(opt_cfg.get("eps", 1e-8))
    tw = [load_audio(u.path, u.start, u.end) for u in train_u]
    tt = [vocab.encode(t) for t in texts]
    vw = [load_audio(u.path, u.start, u.end) for u in valid_u]
    vt = [vocab.encode(remove_special_words(u.text)) for u in valid_u]
    bs, tbs = int(hp["batch_size"]), int(hp.get("test_batch_size", 8))
    eval_steps, num_epochs = int(hp.get("eval_steps", 6400)), int(hp.get("num_epochs", 10))
    sorting = hp.get("sorting", "random")
    if rank == 0:
        os.makedirs(save_dir, exist_ok=True)
        with open(os.path.join(save_dir, "vocab.json"), "w") as f:
            json.dump(vocab.symbols, f, ensure_ascii=False)

    def batch_of(idx, waves, toks):
        x, lens = pad_waves([waves[i] for i in idx])
        t, tl = pad_tokens([toks[i] for i in idx])
        return torch.from_numpy(x), torch.from_numpy(lens.astype(np.float32) / x.shape[1]), torch.from_numpy(t), torch.from_numpy(tl)

    def validate():
        order = sorted(range(len(vw)), key=lambda i: len(vw[i]))  # validation sorted by duration (:417)
        tot, n = 0.0, 0
        for idx in batches_of(order, tbs):
            loss = brain.evaluate_batch(*batch_of(idx, vw, vt), VALID)
            tot, n = tot + float(loss.item()) * len(idx), n + len(idx)
        return tot / max(n, 1)

    # resume (:625-640)
    state = {"step": 0, "epoch": 1, "total_samples": 0, "total_tokens": 0, "total_frames": 0, "train_time_h": 0.0, "valid_time_h": 0.0}
    cks = _ckpt_dirs(save_dir)
    if cks:
        state.update(load_checkpoint(cks[-1], brain))
        if rank == 0:
            print(f"resuming from {cks[-1]} (step {state['step']})")
    step, t0 = 0, time.time()
    run_loss, run_n = 0.0, 0
    for epoch in range(1, num_epochs + 1):
        if sorting == "ascending":
            order = sorted(range(len(tw)), key=lambda i: len(tw[i]))
        elif sorting == "descending":
            order = sorted(range(len(tw)), key=lambda i: -len(tw[i]))
        elif sorting == "random":
            order = list(rng.permutation(len(tw)))
        else:
            raise NotImplementedError("sorting must be random, ascending or descending")
        plan = batches_of(order, bs * world)
        for gi, gidx in enumerate(plan):
            step += 1
            if step <= state["step"]:
                continue  # consumed before the checkpoint this run resumed from
            # every rank takes the SAME number of utterances (data.shard_batch's rule: len // world each, remainder dropped):
            # a tail batch smaller than the world is skipped by ALL ranks, so the collective counts never diverge
            per = len(gidx) // world
            if per == 0:
                continue
            idx = gidx[rank:per * world:world] if world > 1 else gidx
            wavs, wl, toks, tl = batch_of(idx, tw, tt)
            loss = brain.fit_batch(wavs, wl, toks, tl)
            run_loss, run_n = run_loss + float(loss.item()), run_n + 1
            state["total_samples"] += len(gidx)
            state["total_frames"] += int(sum(len(tw[i]) for i in gidx))
            state["total_tokens"] += int(sum(len(tt[i]) for i in gidx))
            last = gi == len(plan) - 1
            if step % eval_steps == 0 or last:
                tv = time.time()
                vloss = validate()
                stats = brain.on_stage_end(VALID, vloss)
                state["valid_time_h"] += (time.time() - tv) / 3600.0
                state["train_time_h"] += (tv - t0) / 3600.0
                t0 = time.time()
                meta = dict(state, step=step, epoch=epoch, WER=float(stats.get("WER", 0.0)), loss=vloss)
                if rank == 0:
                    line = (f"epoch: {epoch}, epoch_finished: {last}, total_samples: {state['total_samples']}, "
                            f"total_audio_h: {state['total_frames'] / (16000 * 3600.0):.6f}, total_tokens: {state['total_tokens']}, "
                            f"train_time_h: {state['train_time_h']:.6f}, valid_time_h: {state['valid_time_h']:.6f}, "
                            f"lr_model: {stats['lr_model']:.3g}, lr_wav2vec: {stats['lr_wav2vec']:.3g} - train loss: {run_loss / max(run_n, 1):.4f} "
                            f"- valid loss: {vloss:.4f}, valid WER: {meta['WER']:.2f}")
                    with open(os.path.join(out_dir, "train_log.txt"), "a") as f:
                        f.write(line + "\n")
                    print(line)
                    save_checkpoint(save_dir, step, brain, meta)
                run_loss, run_n = 0.0, 0
                state["step"] = step
    if rank == 0:
        cks = _ckpt_dirs(save_dir)
        if cks:  # finalize_folder (wav2vec_finalize.py:14-147): final/ = the best checkpoint by WER
            best = min(cks, key=lambda c: json.load(open(os.path.join(c, "meta.json")))["WER"])
            shutil.rmtree(os.path.join(out_dir, "final"), ignore_errors=True)
            shutil.copytree(best, os.path.join(out_dir, "final"))
            shutil.copy(os.path.join(save_dir, "vocab.json"), os.path.join(out_dir, "final", "vocab.json"))
        print(f"trained {step} steps -> {out_dir}")
    if world > 1:
        torch.distributed.destroy_process_group()
    return out_dir


if __name__ == "__main__":
    main()
