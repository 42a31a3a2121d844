"""Training entry point: the counterpart of ``ssak/train/transformers/wav2vec_train.py`` with the train step on
HIP kernels.  Same positional arguments and flags (wav2vec_train.py:144-175); HF ``Trainer`` is replaced by
``ssak_amd.trainer`` (one process per GPU; launch with ``python -m torch.distributed.run --nproc-per-node N
-m ssak_amd.train ...`` for data parallelism over RCCL).  Output folder naming follows :210-243; evaluation
computes eval_loss and WER with greedy decoding every ``--eval_steps`` and writes ``trainer_state.json``.

What decides the output model, as in the reference (wav2vec_train.py:368-372,392,419-420): ``metric_for_best_model="wer"``
(lower is better) is tracked across evaluations, ``EarlyStoppingCallback(early_stopping_patience=15)`` ends the run after 15
evaluations without improvement, ``save_total_limit=2`` keeps the best and the newest checkpoint, and
``load_best_model_at_end`` reloads the best checkpoint's weights before ``final/`` is written.
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import shutil
import time

import numpy as np
import torch

from . import hip
from .checkpoint import load_pretrained, save_pretrained
from .data import (length_grouped_batches, load_audio, load_kaldi, pad_labels, pad_waves, remove_special_words,
                   shard_batch)
from .naming import train_folder_name
from .trainer import AdamW, Trainer


def word_error_rate(refs, hyps) -> float:
    """Word-level Levenshtein distance / reference words (the "wer" metric of wav2vec_train.py:107-125)."""
    errs = tot = 0
    for r, h in zip(refs, hyps):
        r, h = r.split(), h.split()
        d = list(range(len(h) + 1))
        for i, rw in enumerate(r, 1):
            prev, d[0] = d[0], i
            for j, hw in enumerate(h, 1):
                cur = min(d[j] + 1, d[j - 1] + 1, prev + (rw != hw))
                prev, d[j] = d[j], cur
        errs += d[len(h)]
        tot += len(r)
    return errs / max(tot, 1)


EARLY_STOPPING_PATIENCE = 15  # transformers.EarlyStoppingCallback(early_stopping_patience=15), wav2vec_train.py:392


class BestModelTracker:
    """``metric_for_best_model="wer"``, ``greater_is_better=False``, ``load_best_model_at_end=True`` plus
    ``EarlyStoppingCallback(patience)`` as HF Trainer sequences them after every evaluation
    (docker/transformers_modified/trainer.py:2224-2238 and transformers/trainer_callback.py ``check_metric_value``):
    the callback compares the new WER with the best SO FAR (strictly lower resets its counter, anything else -- a tie
    included -- counts as no improvement), THEN the checkpoint is saved and the best metric / checkpoint are updated
    (strictly lower, or nothing recorded yet)."""

    def __init__(self, state: dict, patience: int | None = None):
        self.state, self.patience = state, (EARLY_STOPPING_PATIENCE if patience is None else patience)
        state.setdefault("best_metric", None)
        state.setdefault("best_model_checkpoint", None)
        state.setdefault("early_stopping_patience_counter", 0)

    def after_evaluation(self, wer: float, checkpoint_dir: str) -> bool:
        """Record one evaluation whose checkpoint is ``checkpoint_dir``; True = stop training."""
        st = self.state
        best = st["best_metric"]
        if best is None or wer < best:
            st["early_stopping_patience_counter"] = 0
        else:
            st["early_stopping_patience_counter"] += 1
        if best is None or st["best_model_checkpoint"] is None or wer < best:
            st["best_metric"], st["best_model_checkpoint"] = wer, checkpoint_dir
        return st["early_stopping_patience_counter"] >= self.patience


def rotate_checkpoints(out_dir: str, best: str | None, limit: int = 2):
    """``save_total_limit=2`` (wav2vec_train.py:370) with HF's ordering (trainer.py:2735-2754): oldest first, but the best
    checkpoint is moved up to the second-newest place, so the best and the newest survive."""
    cks = sorted_checkpoints(out_dir)
    if best is not None and os.path.abspath(best) in [os.path.abspath(c) for c in cks]:
        i = [os.path.abspath(c) for c in cks].index(os.path.abspath(best))
        for j in range(i, len(cks) - 2):
            cks[j], cks[j + 1] = cks[j + 1], cks[j]
    for old in cks[:max(0, len(cks) - limit)]:
        shutil.rmtree(old, ignore_errors=True)


def build_parser():
    p = argparse.ArgumentParser(description="Train wav2vec2 (CTC) on Kaldi folders, MI355X HIP path",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("train", help="A kaldi folder, or a file containing a list of kaldi folders, with training data")
    p.add_argument("valid", help="A kaldi folder, or a file containing a list of kaldi folders, with validation data")
    p.add_argument("--debug", default=False, action="store_true")
    p.add_argument("--gpus", default=None)
    p.add_argument("--online", default=False, action="store_true")
    p.add_argument("--max_duration", default=15, type=int)
    p.add_argument("--min_duration", default=1, type=int)
    p.add_argument("--base_model", required=True, type=str, help="Model folder to adapt (HF layout)")
    p.add_argument("--no_freeze", default=False, action="store_true")
    p.add_argument("--data_augment", default=False, action="store_true")
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--batch_size", type=int, default=8)
    p.add_argument("--num_epochs", type=int, default=20)
    p.add_argument("--weight_decay", type=float, default=0.0)
    p.add_argument("--attention_dropout", default=0.1, type=float)
    p.add_argument("--hidden_dropout", default=0.05, type=float)
    p.add_argument("--feat_proj_dropout", default=0.0, type=float)
    p.add_argument("--layer_dropout", default=0.1, type=float)
    p.add_argument("--mask_time_prob", default=0.05, type=float)
    p.add_argument("--disable_first_eval", default=False, action="store_true")
    p.add_argument("--seed", default=69, type=int)
    p.add_argument("--eval_steps", default=400, type=int)
    p.add_argument("--data_augment_noise", default="", type=str, help="(used only with --data_augment)")
    p.add_argument("--data_augment_rir", default="", type=str, help="(used only with --data_augment)")
    p.add_argument("--output_dir", default=".", type=str)
    # extensions (not flags of the reference; declared last and left out of the output-folder name unless used)
    p.add_argument("--batch_audio", type=float, default=None,
                   help="(extension, not a flag of the reference) seconds of PADDED audio per step: length-grouped batches of a constant "
                        "padded length instead of a constant count (many short utterances or few long ones per step).  The number of "
                        "steps per epoch then depends on each epoch's shuffle: the LR-schedule horizon and --num_epochs use the step "
                        "count of ONE seeded plan (an approximation), the logged `epoch` counts utterances actually consumed")
    p.add_argument("--batch_audio_max_utts", type=int, default=None,
                   help="(extension) with --batch_audio: most utterances in one step (default 8 x --batch_size); bounds the workspace "
                        "a step of many short utterances asks for")
    p.add_argument("--skip_unused_layers", default=False, action="store_true",
                   help="(extension, not a flag of the reference) leave the weights and AdamW state of an encoder layer that LayerDrop "
                        "skipped in a step untouched (torch >= 2.0 zero_grad(set_to_none=True) semantics on one GPU); default: the skipped "
                        "layer takes the step with a zero gradient")
    return p


def prepare(utts, tok):
    waves = [load_audio(u.path, u.start, u.end) for u in utts]
    labels = [tok.encode(remove_special_words(u.text)) for u in utts]
    return waves, labels


def evaluate(model, tok, waves, labels, batch_size, rank: int = 0, world: int = 1):
    """eval_loss and eval_wer over the validation set.  Logits never leave the device: greedy decode and the word-level
    edit counts are kernels (ssak_amd.metrics), the loss is summed on the device, and there is ONE host read at the end
    (the reference's compute_metrics argmaxes the full logits on the host at every eval step, wav2vec_train.py:110-125).

    With a process group every rank evaluates ITS contiguous shard of each validation batch -- the reference's
    per_device_eval_batch_size = batch_size // num_devices (wav2vec_train.py:357) -- and ONE all-reduce of four sums
    (word edits, reference words, summed per-utterance loss, utterances) gives every rank the same metrics: no rank idles
    at a barrier while rank 0 walks the whole set.  A shard is padded to the longest utterance of the GLOBAL batch, as the
    reference's collator pads before DataParallel scatters (wav2vec_train.py:79-100): the group-norm base model runs without an
    attention mask, so its logits depend on the padding, and eval_loss / eval_wer (hence best-checkpoint selection and early
    stopping) must not depend on the world size."""
    from .metrics import WerAccumulator
    model.eval()
    acc = WerAccumulator(tok.vocab, tok.pad_token_id, model.device, tok.delim)
    tot = torch.zeros(1, dtype=torch.float64, device=model.device)
    n = 0
    for i in range(0, len(waves), batch_size):
        mine = list(range(i, min(i + batch_size, len(waves))))
        global_len = max(len(waves[k]) for k in mine)
        if world > 1:
            mine = shard_batch(mine, rank, world)
            if not mine:
                continue
        x, lens = pad_waves([waves[k] for k in mine], target_len=global_len)
        lab = torch.from_numpy(pad_labels([labels[k] for k in mine])).to(model.device)
        xd, ld = torch.from_numpy(x).to(model.device), torch.from_numpy(lens).to(model.device)
        use_mask = model.config.feat_extract_norm == "layer"
        fl = torch.tensor([model.num_frames(int(l)) for l in lens], dtype=torch.int32, device=model.device)
        with torch.cuda.device(model.device):
            xn = hip.wave_normalize(xd, ld)
            out = model(xn, lengths=ld if use_mask else None, labels=lab)
            tot += out.loss.double() * len(x)
            acc.add(out.logits, lab, fl)
        n += len(x)
    model.train()
    sums = torch.cat([acc.sums.double(), tot, torch.tensor([float(n)], dtype=torch.float64, device=model.device)])
    if world > 1:
        torch.distributed.all_reduce(sums)  # (exact: integer counts far below 2^53)
    edits, words, loss_sum, count = (float(v) for v in sums.cpu())
    return {"eval_loss": loss_sum / max(count, 1.0), "eval_wer": edits / max(1.0, words)}


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.data_augment:
        raise NotImplementedError("--data_augment (CPU DSP: audiomentations / RIR) is outside the HIP path")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(1, torch.cuda.device_count())  # (rehearsals put several ranks on one card)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI; SSAK_DIST_BACKEND=gloo lets a multi-rank run be rehearsed on a single card (tests)
        backend = os.environ.get("SSAK_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            torch.distributed.init_process_group(backend, rank=rank, world_size=world)
    args.online = args.online or args.data_augment  # wav2vec_train.py:188
    train_u = load_kaldi(args.train, args.min_duration, args.max_duration)
    valid_u = load_kaldi(args.valid, args.min_duration, args.max_duration)
    if args.debug:
        # wav2vec_train.py:279-290 with dataset.py:278-283: the 2 * batch_size LONGEST utterances (does everything fit?)
        keep = 2 * args.batch_size
        train_u = sorted(train_u, key=lambda u: u.duration)[-keep:]
        valid_u = sorted(valid_u, key=lambda u: u.duration)[-keep:]
    else:
        valid_u = valid_u[:480]  # max_data of the validation set, wav2vec_train.py:289
    if len(train_u) < world:
        raise RuntimeError(f"{len(train_u)} training utterances for {world} ranks: nothing to shard")
    script = os.path.abspath(__file__)
    # (the folder names are the reference's, built from ITS options in declaration order: the extension --batch_audio is not one
    # of them and only adds a suffix when it is used)
    named = {k: v for k, v in vars(args).items() if k not in ("batch_audio", "batch_audio_max_utts", "skip_unused_layers")}
    out_dir = os.path.join(args.output_dir, train_folder_name(named, script) + ("" if args.batch_audio is None else f"_ba-{args.batch_audio:g}") + ("_skipunused" if args.skip_unused_layers else ""))
    untrained_dir = os.path.join(args.output_dir, train_folder_name(named, script, untrained=True))
    model, tok = load_pretrained(args.base_model, device=dev, freeze_feature_encoder=not args.no_freeze,
                                 attention_dropout=args.attention_dropout,
                                 hidden_dropout=args.hidden_dropout, feat_proj_dropout=args.feat_proj_dropout,
                                 mask_time_prob=args.mask_time_prob, layerdrop=args.layer_dropout,
                                 ctc_loss_reduction="mean", ctc_zero_infinity=True, pad_token_id=tok_pad(args.base_model))
    model.train()
    # --online (wav2vec_train.py:148: audio loaded on the fly instead of up front): file reads on a background thread, PCM
    # decode / mono mix / resampling / normalisation on the device (ssak_amd.ingest)
    if args.online:
        from .ingest import BatchPrefetcher, DeviceIngest
        ingest = DeviceIngest(16000, dev)
        tw = None
        tl = [tok.encode(remove_special_words(u.text)) for u in train_u]
        train_len = [int(u.duration * 16000) for u in train_u]
    else:
        tw, tl = prepare(train_u, tok)
        train_len = [len(w) for w in tw]
    vw, vl = prepare(valid_u, tok)
    steps_per_epoch = max(1, -(-len(tl) // args.batch_size))  # dataloader_drop_last=False: the short last batch is a step
    total = round(args.num_epochs * len(tl) / args.batch_size)
    if args.batch_audio is not None:  # constant padded length per step: the step count of an epoch follows from the plan
        steps_per_epoch = max(1, len(length_grouped_batches(train_len, args.batch_size, np.random.RandomState(args.seed),
                                                            frame_budget=args.batch_audio * 16000, max_batch=args.batch_audio_max_utts)))
        total = round(args.num_epochs * steps_per_epoch)
    opt = AdamW(model, lr=args.learning_rate, weight_decay=args.weight_decay, warmup_steps=500, total_steps=max(total, 1),
                skip_unused_layers=args.skip_unused_layers)
    trainer = Trainer(model, opt)
    trainer.broadcast_parameters()
    state = {"log_history": [], "global_step": 0, "max_steps": total}
    tracker = BestModelTracker(state)
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "README.txt"), "a") as readme:  # wav2vec_train.py:247-254
            import sys
            print(datetime.datetime.now(), file=readme)
            print(" ".join(sys.argv), file=readme)
            print(f"{len(train_u)} training / {len(valid_u)} validation utterances\n", file=readme)
    # initial evaluation: computed once per (data, base model) in the "untrained" folder, copied into every run's folder,
    # never redone on resume (wav2vec_train.py:397-410).  Rank 0 decides from the files, every rank evaluates its shard.
    init_results = os.path.join(out_dir, "init_eval.json")
    cached = os.path.join(untrained_dir, "init_eval.json")
    need = torch.tensor([int(rank == 0 and not args.disable_first_eval and not os.path.isfile(init_results) and not os.path.exists(cached))],
                        device=dev)
    if world > 1:
        torch.distributed.broadcast(need, src=0)
    if bool(need.item()):
        first = evaluate(model, tok, vw, vl, args.batch_size, rank, world)
        if rank == 0:
            os.makedirs(untrained_dir, exist_ok=True)
            with open(cached, "w") as f:
                json.dump(first, f, indent=1)
    if rank == 0 and not args.disable_first_eval and not os.path.isfile(init_results):
        shutil.copy(cached, init_results)
    rng = np.random.RandomState(args.seed)
    use_mask = model.config.feat_extract_norm == "layer"
    step, t0, run_loss = 0, time.time(), []
    # resume from the last checkpoint of this output folder (wav2vec_train.py:245 get_last_checkpoint -> :415
    # trainer.train(resume_from_checkpoint=...)): weights, optimizer moments and step, the dropout / SpecAugment streams; the
    # batch order is replayed from the seed up to the checkpoint's step
    resume_step = 0
    last = last_checkpoint(out_dir)
    if last is not None:
        from .checkpoint import load_state_dict_file
        model.load_state_dict(load_state_dict_file(last))
        opt.load_state_dict(torch.load(os.path.join(last, "optimizer.pt")))
        with open(os.path.join(last, "trainer_state.json")) as f:
            state = json.load(f)
        tracker = BestModelTracker(state)
        # every rank restores ITS OWN regulariser streams (the trainer seeds ranks differently: seed + rank).  A checkpoint written
        # by fewer ranks has no file for the extra ones: they take rank 0's, then an older build's rng.json; with neither the
        # freshly seeded (seed + rank) streams stay (a warning, not an error: the other ranks would hang at the next collective)
        rng_file = next((f for f in (os.path.join(last, f"rng-rank{rank}.json"), os.path.join(last, "rng-rank0.json"),
                                     os.path.join(last, "rng.json")) if os.path.exists(f)), None)
        if rng_file is None:
            print(f"warning: {last} holds no regulariser stream state for rank {rank}: keeping the freshly seeded streams")
        else:
            with open(rng_file) as f:  # plain JSON, no pickle
                extra = json.load(f)
            if rank > 0 and not rng_file.endswith(f"rng-rank{rank}.json"):
                # the checkpoint was written by fewer ranks: ANOTHER rank's streams would make this rank draw the same dropout /
                # LayerDrop / SpecAugment decisions as that rank for the rest of the run.  Keep the per-rank independence the
                # trainer set up (seed + rank): the step-seed stream is offset by the rank, the host stream reseeded from
                # (seed, rank, step).
                print(f"warning: {last} holds no regulariser stream state for rank {rank} (written by fewer ranks): reseeding this rank's "
                      f"streams from (seed {args.seed}, rank {rank}, step {int(state['global_step'])})")
                model._step_seed = (int(extra["step_seed"]) + 0x9E3779B97F4A7C15 * rank) % (1 << 64)
                model._host_rng = np.random.RandomState([args.seed % (1 << 32), rank, int(state["global_step"])])
            else:
                model._step_seed = int(extra["step_seed"])
                kind, keys, pos, has_gauss, cached_g = extra["host_rng"]
                model._host_rng.set_state((kind, np.asarray(keys, dtype=np.uint32), int(pos), int(has_gauss), float(cached_g)))
        best = state.get("best_model_checkpoint")
        if best is not None and not os.path.isdir(best):  # the output folder was moved: checkpoints are found by name
            state["best_model_checkpoint"] = os.path.join(out_dir, os.path.basename(best))
        resume_step = int(state["global_step"])
        if rank == 0:
            print(f"resuming from {last} (step {resume_step} of {total})")
    seen_utts = 0
    while step < total:
        # (global batch, this rank's contiguous shard of it): shards may differ by one utterance on the short last batch and
        # may be empty; the trainer weights by utterance count
        plan = [(idx, shard_batch(idx, rank, world) if world > 1 else idx) for idx in length_grouped_batches(train_len, args.batch_size, rng,
                                                                frame_budget=None if args.batch_audio is None else args.batch_audio * 16000,
                                                                max_batch=args.batch_audio_max_utts)]
        plan = [gm for gm in plan if gm[0]][:total - step]
        if not plan:
            raise RuntimeError("empty batch plan: no training utterances")
        if step < resume_step:  # batches the checkpointed run already consumed
            skip = min(len(plan), resume_step - step)
            seen_utts += sum(len(w) for w, _ in plan[:skip])
            plan, step = plan[skip:], step + skip
            if not plan:
                continue
        if args.online:
            feed = BatchPrefetcher(ingest, [[(train_u[i].path, train_u[i].start or None, train_u[i].end or None) for i in m] for _, m in plan if m],
                                   labels=[pad_labels([tl[i] for i in m]) for _, m in plan if m])  # (the labels ride with the audio's H2D copy)
            feed = iter(feed)
        else:
            feed = (pad_waves([tw[i] for i in m]) for _, m in plan if m)
        stop = False
        for whole, mine in plan:
            gc = len(whole) if world > 1 else None
            if not mine:  # this rank's shard of a short last batch is empty: zeros into the same collectives
                loss = trainer.train_step(None, None, None, global_count=gc)
            else:
                if args.online:  # already on the device and normalised, labels included
                    x, lens, lab_d = next(feed)
                    loss = trainer.train_step(x, lens, lab_d, raw=False, global_count=gc)
                else:
                    x, lens = next(feed)
                    lab = pad_labels([tl[i] for i in mine])
                    # an un-padded batch (every utterance as long as the batch) carries no length information: lengths=None lets the
                    # trainer fold the normalisation into conv0 (an all-ones mask and no mask are the same computation)
                    full = bool((lens == x.shape[1]).all())
                    loss = trainer.train_step(torch.from_numpy(x).to(dev), None if full else torch.from_numpy(lens).to(dev),
                                              torch.from_numpy(lab).to(dev), global_count=gc)
            run_loss.append(loss)
            step += 1
            seen_utts += len(whole)
            if step % args.eval_steps == 0 or step == total:
                # (frame-budget batches: an epoch is len(tl) utterances consumed, whatever the number of steps that took)
                entry = {"epoch": step / steps_per_epoch if args.batch_audio is None else seen_utts / max(1, len(tl)), "step": step, "learning_rate": opt.current_lr(),
                         "loss": float(torch.stack(run_loss).mean().item())}
                run_loss = []
                ck = os.path.join(out_dir, f"checkpoint-{step}")
                trainer.drain_exchange()  # (exchange "c": its private communicator is idle before torch.distributed's is used)
                metrics = evaluate(model, tok, vw, vl, args.batch_size, rank, world)  # every rank: its shard of each batch
                if rank == 0:
                    state["log_history"].append(entry)
                    state["log_history"].append({"epoch": entry["epoch"], "step": step, **metrics})
                    state["global_step"] = step
                    # on_evaluate (early stopping) first, then save + best-metric bookkeeping, then rotation: HF's order
                    stop = tracker.after_evaluation(metrics["eval_wer"], ck)
                    save_pretrained(model, tok, ck)
                    torch.save(opt.state_dict(), os.path.join(ck, "optimizer.pt"))
                if world > 1:
                    torch.distributed.barrier()  # the checkpoint folder exists
                kind, keys, pos, has_gauss, cached = model._host_rng.get_state()
                with open(os.path.join(ck, f"rng-rank{rank}.json"), "w") as f:  # each rank's own regulariser streams
                    json.dump({"step_seed": int(model._step_seed),
                               "host_rng": [kind, [int(k) for k in keys], int(pos), int(has_gauss), float(cached)]}, f)
                if world > 1:
                    torch.distributed.barrier()  # ... and holds every rank's file before it counts as a checkpoint
                if rank == 0:
                    with open(os.path.join(ck, "trainer_state.json"), "w") as f:
                        json.dump(state, f, indent=1)
                    rotate_checkpoints(out_dir, state["best_model_checkpoint"])
                if world > 1:  # every rank leaves the loop together
                    flag = torch.tensor([int(stop)], device=dev)
                    torch.distributed.broadcast(flag, src=0)
                    stop = bool(flag.item())
                if stop:
                    break
        if stop:
            if rank == 0:
                print(f"early stopping at step {step}: {EARLY_STOPPING_PATIENCE} evaluations without a better WER "
                      f"(best {state['best_metric']:.4f} at {state['best_model_checkpoint']})")
            break
    if rank == 0:
        # load_best_model_at_end (wav2vec_train.py:369; trainer.py:1893-1896): final/ holds the best-WER checkpoint's weights
        best = state.get("best_model_checkpoint")
        if best is not None and os.path.isdir(best):
            from .checkpoint import load_state_dict_file
            model.load_state_dict(load_state_dict_file(best))
            print(f"loading best model from {best} (wer {state['best_metric']:.4f})")
        save_pretrained(model, tok, os.path.join(out_dir, "final"))
        print(f"trained {step} steps in {time.time() - t0:.1f} s -> {out_dir}")
    trainer.close()
    if world > 1:
        torch.distributed.destroy_process_group()


def sorted_checkpoints(out_dir: str):
    if not os.path.isdir(out_dir):
        return []
    cks = [d for d in os.listdir(out_dir) if d.startswith("checkpoint-") and d[11:].isdigit()
           and os.path.exists(os.path.join(out_dir, d, "trainer_state.json"))]
    return [os.path.join(out_dir, d) for d in sorted(cks, key=lambda d: int(d[11:]))]


def last_checkpoint(out_dir: str):
    cks = sorted_checkpoints(out_dir)
    return cks[-1] if cks else None


def tok_pad(folder: str) -> int:
    with open(os.path.join(folder, "vocab.json")) as f:
        return json.load(f).get("<pad>", 0)


if __name__ == "__main__":
    main()
