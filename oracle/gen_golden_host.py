"""String / path fixtures of the host-side helpers, made by IMPORTING the reference.  TEST INFRASTRUCTURE ONLY.

Run in the BUILD container (``python -m oracle.gen_golden_host``): ``ssak.utils.text_basic``, ``ssak.utils.misc`` and
``ssak.utils.train_utils`` import here from /root/reference (the modules of the hot path itself do
not: ordinary ModuleNotFoundError).  Writes ``tests/golden/host_strings.json``:

* ``remove_special_words``  (ssak/utils/text_basic.py:91-125)   text -> cleaned label text, as ``process_dataset`` applies it;
* ``hashmd5``               (ssak/utils/misc.py:42-46)           object -> md5 of its pickle (output-folder names);
* ``remove_commonprefix``   (ssak/utils/misc.py:76-92);
* ``args_to_str``           (ssak/utils/train_utils.py:4-16)     generic hyper-parameter string;
* ``output_dir``            the output-folder names the reference's own test asserts
                            (tests/unittests/test_train_transformers.py:23-24,55-56: golden strings, read from that file)
                            together with the command line that must produce them (wav2vec_train.py:210-239);
* ``wavscp``                the lines of the reference's test Kaldi folders (tests/data/kaldi/*/wav.scp: data) and the audio
                            path each must resolve to under ``parse_kaldi_wavscp`` (ssak/utils/kaldi.py:8-37), plus the list
                            file tests/data/kaldi/train_weighted.txt.
"""
from __future__ import annotations

import json
import os
import re
import sys

REF = "/root/reference"
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _spacing_tables(src: str):
    """lang_spec_sub / default_sub of tools/align_audio_transcript.py:34-53 evaluated from the file's text (data tables)."""
    ns = {"re": re}
    a = src.index("lang_spec_sub = {")
    b = src.index("DEFAULT_ALIGN_MODELS_TORCH")
    exec(src[a:b], ns)
    d = dict(ns["lang_spec_sub"])
    d["default"] = ns["default_sub"]
    return d


def main():
    sys.path.insert(0, REF)
    from ssak.utils.misc import hashmd5, remove_commonprefix
    from ssak.utils.text_basic import remove_special_words
    from ssak.utils.train_utils import args_to_str
    out = {}
    texts = ["bonjour <noise> tout le monde", "<noise>", "", "l ' amour  de   l' art", "aujourd ' hui <laugh> c ' est <b> bien",
             "a<x>b <y> c", "rock ' n ' roll", "peut - être", "  espaces   multiples  ", "d'accord", "<unk> euh <unk>",
             "tab\there\nnewline", "it ' s <a b c> fine ' ok"]
    out["remove_special_words"] = [[t, {}, remove_special_words(t)] for t in texts]
    out["remove_special_words"] += [[t, {"glue_apostrophe": False}, remove_special_words(t, glue_apostrophe=False)] for t in texts[:8]]
    out["remove_special_words"] += [[t, {"glue_apostrophe": None}, remove_special_words(t, glue_apostrophe=None)] for t in texts[:8]]
    objs = [["tests/data/kaldi/train_weighted.txt", "tests/data/kaldi/minimal"], ["a", "b"], ["/x/y", "/x/z"], ["", ""]]
    out["hashmd5_tuple"] = [[o, hashmd5(tuple(o))] for o in objs]
    lists = [["/a/b/ssak/train/x.py", "/a/b/tests/data/kaldi/t.txt", "/a/b/tests/data/kaldi/minimal"], ["/a/bc/d", "/a/bd/e"],
             ["/same/path", "/same/path"], ["x", "y"]]
    out["remove_commonprefix"] = [[l, "/", remove_commonprefix(list(l), "/")] for l in lists]
    dicts = [dict(learning_rate=1e-4, batch_size=8, seed=69, gpus="0", lr=0.5), dict(max_duration=15, freeze=True, x=None, name="a/b"),
             dict(weight_decay=0.0, warmup_steps=500, use_peft=False)]
    out["args_to_str"] = [[d, args_to_str(dict(d))] for d in dicts]
    out["args_to_str_sorted"] = [[d, args_to_str(dict(d), sort=True)] for d in dicts]
    # output-folder names asserted by the reference's own test (golden strings live in the test file)
    src = open(os.path.join(REF, "tests/unittests/test_train_transformers.py")).read()
    dir0 = re.findall(r'dir0 = self.get_output_path\("([^"]+)"\)', src)
    tails = re.findall(r'dir = dir0 \+ "([^"]+)"', src)
    assert len(dir0) == 2 and len(tails) == 2 and dir0[0] == dir0[1]
    md5 = hashmd5(tuple(remove_commonprefix([REF + "/ssak/train/transformers/wav2vec_train.py", REF + "/tests/data/kaldi/train_weighted.txt",
                                             REF + "/tests/data/kaldi/minimal"], "/")[1:]))
    assert dir0[0].startswith("hf_" + md5 + "_"), (dir0[0], md5)
    common = ["--batch_size", "4", "--num_epochs", "1", "--eval_steps", "1", "--max_duration", "10"]
    out["output_dir"] = [
        dict(script="ssak/train/transformers/wav2vec_train.py", train="tests/data/kaldi/train_weighted.txt", valid="tests/data/kaldi/minimal",
             flags=common, base_model="Ilyes/wav2vec2-large-xlsr-53-french", untrained=dir0[0], output=dir0[0] + tails[0]),
        dict(script="ssak/train/transformers/wav2vec_train.py", train="tests/data/kaldi/train_weighted.txt", valid="tests/data/kaldi/minimal",
             flags=common + ["--data_augment", "--data_augment_noise", "tests/data/noise", "--data_augment_rir", "tests/data/[rirs/smallroom/rir_list]"],
             base_model="Ilyes/wav2vec2-large-xlsr-53-french", untrained=dir0[1], output=dir0[1] + tails[1]),
    ]
    # wav.scp lines held by the reference's tests (data) and what each id resolves to
    ws = {}
    for name in ("minimal", "small", "complete"):
        lines = open(os.path.join(REF, "tests/data/kaldi", name, "wav.scp")).read().splitlines()
        exp = {}
        for ln in lines:
            # the two forms these folders use: `id sox PATH -t wav ... |` and the same with a quoted path holding a space
            m = re.match(r"(\S+) sox '([^']+)' -t wav -r 16k -b 16 -c 1 - \|$", ln) or \
                re.match(r"(\S+) sox (\S+) -t wav -r 16k -b 16 -c 1 - \|$", ln)
            assert m, ln
            exp[m.group(1)] = m.group(2)
        ws[name] = dict(lines=lines, expected=exp)
    # text clean-up of the alignment tool (tools/align_audio_transcript.py:78-118) built from the reference's importable pieces;
    # its emoji / numbers-to-words steps (ssak.utils.text_utils: num2words) are not importable and not on this path
    from ssak.utils.text_basic import collapse_whitespace, format_special_characters, remove_punctuations, remove_quotes
    sys.path.insert(0, os.path.join(REF, "tools"))
    lang_sub = {  # the punctuation-spacing tables of tools/align_audio_transcript.py:34-53, read from the file (it cannot be imported: soxbindings)
        k: v for k, v in _spacing_tables(open(os.path.join(REF, "tools/align_audio_transcript.py")).read()).items()}
    strs = ["Bonjour  «le monde» … ça va ?", "l’été – déjà ! Oui:non;peut-être,ok.Fin", "e\u0301te\u0301 a\u0300 la plage", "``quoted'' 'single' \"double\"",
            "<noise> euh , d ' accord <laugh>", "tiret - isolé - fin -", "1ᵉʳ et 2ᵉ · point", "Œuvre æther", "espaces\u00a0insécables\u202fici",
            "What?No!Yes:ok", "fin.Début Autre.suite", "a,b ,c , d", "bonjour", "", "   "]
    cases = []
    for t in strs:
        for lang in ("fr", "en"):
            x = remove_quotes(format_special_characters(t, remove_ligatures=False))
            x = remove_special_words(x)
            for a, b in lang_sub.get(lang, lang_sub["default"]):
                x = re.sub(a, b, x)
            cases.append([t, lang, collapse_whitespace(x)])
    out["align_text_normalization"] = cases
    out["align_word_normalization"] = [[w, lig, pun, (lambda y: (remove_punctuations(y) or y))(format_special_characters(w, remove_ligatures=lig)) if pun
                                        else format_special_characters(w, remove_ligatures=lig)]
                                       for w in ["Œuvre,", "(mot)", "...", "l'été.", "straße", "a-b", "«quoi»", "ok ,"] for lig in (False, True) for pun in (False, True)]
    ws["other_forms"] = dict(lines=["a /data/a.wav", "b flac -c -d -s -f /data/b.flac |", "c sox '/data/with space/c.wav' -t wav - |",
                                    "d /usr/bin/sox $DATAPATH/d.mp3 -t wav -r 16000 -b 16 - |"],
                             expected={"a": "/data/a.wav", "b": "/data/b.flac", "c": "/data/with space/c.wav", "d": "$DATAPATH/d.mp3"})
    out["wavscp"] = ws
    out["list_file"] = open(os.path.join(REF, "tests/data/kaldi/train_weighted.txt")).read()
    with open(os.path.join(GOLD, "host_strings.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print("host_strings ok:", {k: len(v) for k, v in out.items()})


if __name__ == "__main__":
    main()
