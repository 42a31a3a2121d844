#!/usr/bin/env python3
"""Split the long utterances of a Kaldi folder at word boundaries found by CTC forced alignment: the counterpart of the
reference's ``tools/align_audio_transcript.py`` (``split_long_audio_kaldifolder`` :121-335, command line :445-520), the
consumer of ``compute_alignment`` (ssak/utils/align_transcriptions.py:294-402).

Same function name, arguments, output files (``text``, ``utt2spk``, ``utt2dur``, ``segments`` + copies of ``wav.scp`` /
``spk2gender``), cut ids (``<id>_cutNN``), number formats and resume-into-an-existing-folder behaviour.  What differs is the
shape of the work, which is what the device wants: the reference aligns utterance by utterance inside its loop (one Python
frame loop per utterance); here a first pass over the folder PLANS the work -- every utterance becomes either a line to copy
through or an alignment job -- and the jobs are aligned ``batch_size`` at a time with ONE launch of the trellis + backtrack
kernel (``ssak_ctc_forced_align_batch``: one workgroup per utterance, 256 CUs), after which the cut points are decided on the
host exactly as the reference decides them.  Output order is the input order.

Outside this path (raise / documented): plotting (``plot=True``), the debug folder of cut audio (needs sox), the
reference's number-to-words and emoji normalisers (``numbers_and_symbols_to_letters`` / ``remove_special_characters`` need
num2words and an emoji table: text normalisation is out of scope) -- ``word_normalizer`` lets a caller plug
its own -- and Kaldi's ``fix_data_dir.sh`` (sorting and ``spk2utt`` are done here).
"""
from __future__ import annotations

import dataclasses
import logging
import os
import re
import shutil
import unicodedata
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

logger = logging.getLogger(__name__)

# ssak/utils/text_basic.py:15-16: string.punctuation + the listed extra marks, minus "-" and "'"
PUNCTUATION = "".join(c for c in __import__("string").punctuation + "。，！？：”、…" + "؟،؛" + "—" + "«°»×‹›•“–‘″‘" if c not in "-'")
_PUNCT_CLASS = "[" + re.escape(PUNCTUATION) + "]"

# single characters brought to their plain forms (the character part of format_special_characters, text_basic.py:28-82)
_CHAR_MAP = str.maketrans({"\x92": "'", "’": "'", "‘": "'", "‛": "'", "ʿ": "'", "‚": ",", "–": "-", " ": " ", " ": " ",
                           "·": ".", "ᵉ": "e"})
_CONTROL = re.compile(r"[\x00-\x08\x0B\x0C\x0E-\x1F\x7F-\x9F]")

# spacing around punctuation marks (tools/align_audio_transcript.py:34-53); French puts a space before ? ! : ;
_SPACING = {
    "fr": [(r"([?!:;])", r" \1"), (r"\s+([,.])", r"\1"), (r"([?!:;,]+)([^ ?!:;,\d])", r"\1 \2"), (r"([.]+)([A-Z])", r"\1 \2")],
    None: [(r"\s+([?!:;,.])", r"\1"), (r"([?!:;,]+)([^ ?!:;,\d])", r"\1 \2"), (r"([.]+)([A-Z])", r"\1 \2")],
}


def collapse_whitespace(text: str) -> str:
    return re.sub(r"\s+", " ", text).strip()


def format_special_characters(text: str, remove_ligatures: bool = False) -> str:
    """Typographic variants -> plain characters: accents composed (NFC), curly quotes / guillemets -> straight quotes,
    ellipsis -> three dots, isolated dashes dropped (semantics of text_basic.py:28-82)."""
    text = unicodedata.normalize("NFC", text).replace("ᵉʳ", "er").translate(_CHAR_MAP).replace("…", "...")
    text = re.sub(r"[«“][^\S\r\n]*", '"', text)
    text = re.sub(r"[^\S\r\n]*[»”″„]", '"', text)
    text = re.sub(r"(``|'')", '"', text)
    text = _CONTROL.sub("", text)
    if remove_ligatures:
        for a, b in (("œ", "oe"), ("æ", "ae"), ("ﬁ", "fi"), ("ﬂ", "fl"), ("ĳ", "ij"), ("Œ", "Oe"), ("Æ", "Ae")):
            text = text.replace(a, b)
    text = re.sub(" - | -$|^- ", " ", text)
    return collapse_whitespace(text)


def remove_quotes(text: str) -> str:
    text = text.replace('"', "")
    text = re.sub(r"''+", "", text)
    return re.sub(r" '([^']+)'", r" \1", text)


def custom_text_normalization(transcript: str, regex_rm=None, lang: str = "fr") -> str:
    """Utterance-level clean-up before word splitting (tools/align_audio_transcript.py:78-96)."""
    from ..data import remove_special_words
    transcript = remove_quotes(format_special_characters(transcript))
    if regex_rm:
        for rx in ([regex_rm] if isinstance(regex_rm, str) else regex_rm):
            transcript = re.sub(rx, "", transcript)
    else:
        transcript = remove_special_words(transcript)
    for pat, repl in _SPACING.get(lang, _SPACING[None]):
        transcript = re.sub(pat, repl, transcript)
    return collapse_whitespace(transcript)


def labels_to_norm_args(labels: Sequence[str]) -> dict:
    """What the acoustic model cannot spell must be rewritten in the words handed to the aligner (:98-104)."""
    return {"remove_digits": "9" not in labels, "remove_punc": "." not in labels,
            "remove_ligatures": "œ" not in labels and "æ" not in labels, "remove_etset": "ß" not in labels}


def custom_word_normalization(word: str, lang: str, remove_digits: bool, remove_punc: bool, remove_ligatures: bool, remove_etset: bool,
                              digits_to_letters: Optional[Callable[[str, str], str]] = None) -> str:
    """Word-level rewriting that keeps the word segmentation (:106-118).  ``digits_to_letters(word, lang)`` stands for the
    reference's numbers_and_symbols_to_letters (num2words: not on this path); without it digits stay and are aligned as
    word separators by ``loose_get_char_index``."""
    word = format_special_characters(word, remove_ligatures=remove_ligatures)
    if remove_digits and digits_to_letters is not None:
        word = digits_to_letters(word, lang)
    if remove_etset:
        word = word.replace("ß", "ss")
    if remove_punc:
        stripped = re.sub(_PUNCT_CLASS, "", word)
        if stripped:
            word = stripped
    return collapse_whitespace(word)


# ------------------------------------------------------------------------------------------------ the plan
@dataclasses.dataclass
class CopyLine:
    """An utterance that goes through unchanged (short enough, nothing to refine)."""
    id: str
    text: str
    spk: str
    dur: float
    seg: Tuple[str, float, float]


@dataclasses.dataclass
class AlignJob:
    """An utterance to align and cut."""
    id: str
    spk: str
    wavid: str
    path: str
    start: float
    end: float
    dur: float
    original_text: str
    words: List[str]          # as they will be written
    spoken: List[str]         # as they are handed to the aligner (same segmentation)
    first_of_file: bool
    last_of_file: bool
    weird: bool               # its duration was a "up to the next segment" marker


def read_kaldi_inputs(dirin: str, glue_starting_punctuation_to_previous: bool):
    """text / utt2spk / utt2dur / segments / wav.scp of the input folder (:176-236)."""
    from ..data import parse_kaldi_wavscp
    id2text: Dict[str, str] = {}
    prev = None
    with open(os.path.join(dirin, "text")) as f:
        for line in f:
            parts = line.strip().split(" ", 1)
            if len(parts) == 1:
                continue
            uid, text = parts[0], parts[1].strip()
            if (glue_starting_punctuation_to_previous and prev and text and text[0] in ".,:;?!" and (len(text) == 1 or text[1] == " ")
                    and id2text[prev][-1] not in ".,:;?!"):
                id2text[prev] += text[0]  # a line that starts with a mark belongs to the previous utterance
                text = text[1:].strip()
            if not text:
                continue
            id2text[uid] = text
            prev = uid
    with open(os.path.join(dirin, "utt2spk")) as f:
        id2spk = dict(line.strip().split() for line in f if line.strip())
    id2dur: Dict[str, float] = {}
    with open(os.path.join(dirin, "utt2dur")) as f:
        for line in f:
            if line.strip():
                uid, dur = line.strip().split(" ")
                id2dur[uid] = float(dur)
    has_segments = os.path.isfile(os.path.join(dirin, "segments"))
    if has_segments:
        id2seg = {}
        with open(os.path.join(dirin, "segments")) as f:
            for line in f:
                if line.strip():
                    uid, wav, a, b = line.strip().split(" ")
                    id2seg[uid] = (wav, float(a), float(b))
    else:
        id2seg = {uid: (uid, 0, id2dur[uid]) for uid in id2dur}
    wav2path = parse_kaldi_wavscp(os.path.join(dirin, "wav.scp"))
    return id2text, id2spk, id2dur, id2seg, wav2path, has_segments


def plan_folder(ids: Sequence[str], id2text, id2spk, id2dur, id2seg, wav2path, has_segments, labels, *, min_duration, max_duration,
                refine_timestamps, lang, regex_rm_part, regex_rm_full, special_duration_meaning_tonext, can_reject_based_on_score,
                word_normalizer=None):
    """First pass: one CopyLine / AlignJob per kept utterance, in input order (the decisions of :238-331)."""
    norm_args = labels_to_norm_args(labels)
    previous_path = None
    for k, uid in enumerate(ids):
        if uid not in id2text:
            continue  # empty transcription
        dur = id2dur[uid]
        wavid, start, end = id2seg[uid]
        path = wav2path[wavid]
        first_of_file = previous_path != path
        previous_path = path
        original = id2text[uid]
        transcript = custom_text_normalization(original, regex_rm=regex_rm_part, lang=lang)
        if not transcript:
            logger.warning(f'{uid} with transcript "{original}" removed because of empty transcript after normalization.')
            continue
        if regex_rm_full and any(re.search(r"^" + rx + r"$", transcript) for rx in regex_rm_full):
            logger.warning(f'{uid} with transcript "{original}" removed because of a full-utterance regex')
            continue
        nxt = ids[k + 1] if k + 1 < len(ids) else None
        next_path = wav2path[id2seg[nxt][0]] if nxt is not None else None
        weird = False
        # on some sources a tiny duration (0.001, 0.002) means "up to the next segment"
        if refine_timestamps and has_segments and min(abs(dur - d) for d in special_duration_meaning_tonext) < 0.0001 and nxt is not None:
            if path == next_path:
                new_dur = id2seg[nxt][1] - start
                logger.warning(f'changing duration from {dur:.3f} to {new_dur:.3f} for {uid} with transcript "{original}"')
                dur, end, weird = new_dur, start + new_dur, True
                id2seg[uid] = (wavid, start, end)
        if dur <= min_duration:
            logger.warning(f'{uid} with transcript "{original}" removed because of small duration {dur}.')
            continue
        if dur <= max_duration and not refine_timestamps and not can_reject_based_on_score:
            yield CopyLine(uid, transcript, id2spk[uid], id2dur[uid], id2seg[uid])
            continue
        words: List[str] = []
        for w in transcript.split():  # an isolated punctuation mark rides with the word before it
            if words and re.sub(rf"[ {re.escape(PUNCTUATION)}]", "", w) == "":
                words[-1] += " " + w
            else:
                words.append(w)
        norm = word_normalizer or (lambda w: custom_word_normalization(w, lang=lang, **norm_args))
        spoken = [norm(w) for w in words]
        if refine_timestamps:
            start, end = max(0, start - refine_timestamps), end + refine_timestamps  # (the audio loader clips the end)
        yield AlignJob(uid, id2spk[uid], wavid, path, start, end, dur, original, words, spoken, first_of_file,
                       nxt is None or path != next_path, weird)


# ------------------------------------------------------------------------------------------------ cutting
def cut_at_word_boundaries(word_segments, words: Sequence[str], num_frames: int, audio_len: int, sample_rate: int, max_duration: float,
                           refine_timestamps) -> List[Tuple[int, float, float, str]]:
    """The cut decision of the reference (:395-437) as a pure function: walk the aligned words, close a piece whenever the next
    word would carry it past ``max_duration``.  Returns [(cut index from 1, start s, end s, text)] relative to the audio's
    start; pieces of null or negative duration are kept in the list (the writer skips them with a warning, as :360-361)."""
    assert len(word_segments) == len(words), f"{[w.label for w in word_segments]}\n{list(words)}\n{len(word_segments)} != {len(words)}"
    ratio = audio_len / (num_frames * sample_rate)
    segs = [dataclasses.replace(s) for s in word_segments]
    if segs and not refine_timestamps:
        segs[0].start = 0
        segs[-1].end = num_frames
    pieces: List[Tuple[int, float, float, str]] = []
    piece_start = piece_end = 0.0
    text = ""
    for i, (seg, word) in enumerate(zip(segs, words)):
        if word.strip() in PUNCTUATION:  # a punctuation-only word takes no time
            seg.end = seg.start
            if text == "":
                logger.warning("removed a punctuation mark???")
        if refine_timestamps and i == 0:
            piece_start = piece_end = seg.start * ratio
        end = seg.end * ratio
        if end - piece_start > max_duration and text:
            pieces.append((len(pieces) + 1, piece_start, piece_end, text))
            piece_start, text = piece_end, ""
        piece_end = end
        text = (text + " " if text else "") + word
    if text:
        pieces.append((len(pieces) + 1, piece_start, segs[-1].end * ratio, text))
    return pieces


def _last_line(path: str) -> Optional[str]:
    last = None
    with open(path) as f:
        for last in f:
            pass
    return last


def fix_kaldi_dir(dirname: str):
    """What the output folder needs from Kaldi's fix_data_dir.sh here: files sorted by id, spk2utt derived from utt2spk."""
    for name in ("text", "utt2spk", "utt2dur", "segments"):
        p = os.path.join(dirname, name)
        if os.path.isfile(p):
            with open(p) as f:
                lines = sorted(set(l for l in f if l.strip()), key=lambda l: l.split(" ", 1)[0])
            with open(p, "w") as f:
                f.writelines(lines)
    p = os.path.join(dirname, "utt2spk")
    if os.path.isfile(p):
        spk: Dict[str, List[str]] = {}
        with open(p) as f:
            for line in f:
                u, s = line.split()
                spk.setdefault(s, []).append(u)
        with open(os.path.join(dirname, "spk2utt"), "w") as f:
            for s in sorted(spk):
                f.write(s + " " + " ".join(spk[s]) + "\n")


def split_long_audio_kaldifolder(dirin, dirout, model, min_duration=0, max_duration=30, refine_timestamps=None, lang="fr",
                                 regex_rm_part=None, regex_rm_full=None, special_duration_meaning_tonext=(0.001, 0.002),
                                 can_reject_based_on_score=False, can_reject_only_first_and_last=True,
                                 glue_starting_punctuation_to_previous=True, verbose=False, debug_folder=None, plot=False,
                                 skip_warnings=False, batch_size=32, word_normalizer=None):
    """Split long audio files into smaller ones (arguments of tools/align_audio_transcript.py:121-158; ``batch_size`` = utterances
    aligned per kernel launch, ``word_normalizer`` = optional replacement for the word-level rewriting)."""
    from .. import align as A
    from ..data import load_audio
    from ..infer import transformers_load_model
    if plot or debug_folder:
        raise NotImplementedError("plot / debug_folder (sox) are outside the device path")
    assert dirout != dirin
    last_id = None
    if os.path.isdir(dirout):
        for name in ("utt2dur", "text", "utt2spk", "segments"):
            if not os.path.isfile(os.path.join(dirout, name)):
                raise RuntimeError(f"Folder {dirout} already exists but does not contain file {name}. Aborting (remove the folder to retry)")
        logger.warning(f"{dirout} already exists. Continuing with unprocessed.")
        line = _last_line(os.path.join(dirout, "utt2dur"))
        if line:
            last_complete = last_id = line.split()[0]
            if re.match(r".+_cut\d+$", last_complete):
                last_id = "_cut".join(last_complete.split("_cut")[:-1])
            for name in ("text", "utt2spk", "segments"):  # the id must have been written everywhere
                other = _last_line(os.path.join(dirout, name))
                assert other and other.split()[0] == last_complete, f"Last id {last_complete} in utt2dur does not match {name}"
    os.makedirs(dirout, exist_ok=True)
    model = transformers_load_model(model)
    sample_rate = 16000
    labels, _ = A.get_model_vocab(model)
    id2text, id2spk, id2dur, id2seg, wav2path, has_segments = read_kaldi_inputs(dirin, glue_starting_punctuation_to_previous)
    ids = list(id2dur)
    if last_id is not None:
        if last_id not in id2dur:
            raise RuntimeError(f"Last processed id {last_id} not found in {dirin}/utt2dur")
        ids = ids[ids.index(last_id) + 1:]
        if not ids:
            logger.warning(f"{dirout} already exists and is complete. Aborting.")
            return
    plan = plan_folder(ids, id2text, id2spk, id2dur, id2seg, wav2path, has_segments, labels, min_duration=min_duration,
                       max_duration=max_duration, refine_timestamps=refine_timestamps, lang=lang, regex_rm_part=regex_rm_part,
                       regex_rm_full=regex_rm_full, special_duration_meaning_tonext=list(special_duration_meaning_tonext),
                       can_reject_based_on_score=can_reject_based_on_score, word_normalizer=word_normalizer)
    has_shorten = False
    with open(os.path.join(dirout, "text"), "a") as f_text, open(os.path.join(dirout, "utt2spk"), "a") as f_spk, \
            open(os.path.join(dirout, "utt2dur"), "a") as f_dur, open(os.path.join(dirout, "segments"), "a") as f_seg:

        def flush(pending):
            """Align the jobs among ``pending`` in ONE launch, then write every pending item in input order."""
            nonlocal has_shorten
            jobs = [it for it in pending if isinstance(it, AlignJob)]
            audios, kept = [], []
            for j in jobs:
                try:
                    audios.append(load_audio(j.path, j.start, j.end, sample_rate))
                    kept.append(j)
                except RuntimeError as err:
                    logger.warning(f'{j.id} with transcript "{j.original_text}" removed because of audio loading error: {err}')
            results = dict(zip((j.id for j in kept), A.compute_alignment_batch(audios, [j.spoken for j in kept], model,
                                                                             first_as_garbage=bool(refine_timestamps)))) if kept else {}
            alen = {j.id: len(a) for j, a in zip(kept, audios)}
            for it in pending:
                if isinstance(it, CopyLine):
                    f_text.write(f"{it.id} {it.text}\n")
                    f_spk.write(f"{it.id} {it.spk}\n")
                    f_dur.write(f"{it.id} {it.dur}\n")
                    f_seg.write(f"{it.id} {it.seg[0]} {it.seg[1]} {it.seg[2]}\n")
                    continue
                res = results.get(it.id)
                if res is None:
                    continue
                if isinstance(res, Exception):
                    logger.warning(f'{it.id} with transcript "{it.original_text}" removed because of alignment error: {res}')
                    continue
                num_frames, char_segments, word_segments = res
                if can_reject_based_on_score:
                    score = max(np.mean([s.score for s in char_segments]), np.mean([s.score for s in word_segments]))
                    if score < 0.4 and (not can_reject_only_first_and_last or it.weird or it.first_of_file or it.last_of_file):
                        logger.warning(f'{it.id} with transcript "{it.original_text}" removed because of score {score} < 0.4')
                        continue
                has_shorten = True
                for index, a, b, text in cut_at_word_boundaries(word_segments, it.words, num_frames, alen[it.id], sample_rate,
                                                                max_duration, refine_timestamps):
                    new_id = f"{it.id}_cut{index:02}"
                    new_start, new_end = it.start + a, it.start + b
                    too_long = new_end - new_start > max_duration
                    if too_long and not skip_warnings:
                        logger.warning(f"{new_id} got long sequence {new_end - new_start} > {max_duration} (transcript={text})")
                    if b <= a:
                        logger.warning(f"Skipping {new_id}, got null or negative duration (after realignment, start={a}, end={b})")
                    elif too_long and skip_warnings:
                        logger.warning(f"Skipping {new_id} got long sequence {new_end - new_start} > {max_duration}")
                    else:
                        f_text.write(f"{new_id} {text}\n")
                        f_spk.write(f"{new_id} {it.spk}\n")
                        f_dur.write(f"{new_id} {new_end - new_start:.3f}\n")
                        f_seg.write(f"{new_id} {it.wavid} {new_start:.3f} {new_end:.3f}\n")
            for f in (f_text, f_spk, f_dur, f_seg):
                f.flush()

        pending, njobs = [], 0
        for item in plan:
            pending.append(item)
            njobs += isinstance(item, AlignJob)
            if njobs >= batch_size:
                flush(pending)
                pending, njobs = [], 0
        if pending:
            flush(pending)
    if not has_shorten:
        logger.info("No audio was shorten. Folder should be (quasi) unchanged")
        if not has_segments:
            os.remove(os.path.join(dirout, "segments"))
    for name in ("wav.scp", "spk2gender"):
        if os.path.isfile(os.path.join(dirin, name)):
            shutil.copy(os.path.join(dirin, name), os.path.join(dirout, name))
    fix_kaldi_dir(dirout)


def build_parser():
    import argparse
    p = argparse.ArgumentParser(description="Split long annotations into smaller ones", formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("dirin", help="Input folder", type=str)
    g = p.add_mutually_exclusive_group(required=True)
    g.add_argument("--pattern_in", help="If specified, activates batch (process all subfolders of dirin)", type=str, default=None)
    g.add_argument("--dirout", help="Output folder", type=str)
    p.add_argument("--language", default="fr", help="Language (for text normalizations)")
    p.add_argument("--model", help="Acoustic model to align (folder in HF layout)", type=str, required=True)
    p.add_argument("--min_duration", default=0.005, type=float)
    p.add_argument("--max_duration", help="Maximum length (in seconds)", default=30, type=float)
    p.add_argument("--refine_timestamps", help="A value (in seconds) to refine timestamps with", default=None, type=float)
    p.add_argument("--regex_rm_part", type=str, nargs="*", default=["\\[[^\\]]*\\]", "\\([^\\)]*\\)", "<[^>]*>"],
                   help="One or several regex to remove parts from the transcription.")
    p.add_argument("--regex_rm_full", type=str, nargs="*",
                   default=[" *[Vv]idéo sous-titrée par.*", " *SOUS-TITRES.+", " *[Ss]ous-titres.+", " *SOUS-TITRAGE.+", " *[Ss]ous-titrage.+",
                            " *\\.+ *"], help="One or several regex to remove a full utterance.")
    p.add_argument("--gpus", help="List of GPU index to use (starting from 0)", default=None)
    p.add_argument("--debug_folder", default=None, type=str)
    p.add_argument("--plot", default=False, action="store_true")
    p.add_argument("--verbose", default=False, action="store_true")
    p.add_argument("--skip_warnings", default=False, action="store_true", help="If True, it will not keep rows with warnings")
    p.add_argument("--batch_size", default=32, type=int, help="utterances aligned per kernel launch")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO)
    if args.gpus:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", str(args.gpus))
    if args.pattern_in:  # every sub-folder that matches, into a sibling with the same name under <dirin>_split
        targets = [(os.path.join(args.dirin, d), os.path.join(args.dirin.rstrip("/") + "_split", d))
                   for d in sorted(os.listdir(args.dirin)) if re.search(args.pattern_in, d) and os.path.isdir(os.path.join(args.dirin, d))]
    else:
        targets = [(args.dirin, args.dirout)]
    for dirin, dirout in targets:
        split_long_audio_kaldifolder(dirin, dirout, model=args.model, lang=args.language, min_duration=args.min_duration,
                                     max_duration=args.max_duration, refine_timestamps=args.refine_timestamps,
                                     regex_rm_part=args.regex_rm_part, regex_rm_full=args.regex_rm_full, debug_folder=args.debug_folder,
                                     plot=args.plot, verbose=args.verbose, skip_warnings=args.skip_warnings, batch_size=args.batch_size)


if __name__ == "__main__":
    main()
