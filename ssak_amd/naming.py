"""Output-folder naming of the training entry point (host glue, no device work).

The reference encodes the data and the hyper-parameters in the folder name so that a rerun lands in (and resumes from) the
same folder, and its own test asserts the exact strings (ssak/train/transformers/wav2vec_train.py:210-239;
tests/unittests/test_train_transformers.py:23-24,55-56).  Pinned by tests/golden/host_strings.json (made by importing
``ssak.utils.misc`` / ``ssak.utils.train_utils`` from the reference).
"""
from __future__ import annotations

import hashlib
import os
import pickle
from typing import Iterable, List, Mapping, Optional, Sequence


def hashmd5(obj) -> str:
    """md5 of the object's pickle (ssak/utils/misc.py:42-46).  Stable for the str tuples it is used on."""
    return hashlib.md5(pickle.dumps(obj)).hexdigest()


def strip_common_prefix(paths: Sequence[str], stop: Optional[str] = None) -> List[str]:
    """Drop the longest leading string shared by all ``paths``; with ``stop`` the shared part is shortened until it ends
    with that separator (ssak/utils/misc.py:76-92)."""
    if not paths:
        return []
    shared = os.path.commonprefix(list(paths))  # character-wise, like the reference's min/max scan
    if stop:
        cut = shared.rfind(stop)
        shared = shared[:cut + len(stop)] if cut >= 0 else ""
    return [p[len(shared):] for p in paths]


def _initials(key: str) -> str:
    return "".join(word[:1] for word in key.replace("-", "_").split("_"))


def _compact(value) -> str:
    # 0 / 0.0 / False print as 0 and 1 / 1.0 / True as 1 (the reference looks the value up in {True: 1, False: 0})
    if isinstance(value, (bool, int, float)) and value in (0, 1):
        return str(int(value))
    return str(value).replace("/", "_")


# options that do not enter the name: without influence on the result, or handled by a suffix / the data hash
_NOT_NAMED = frozenset(("verbose", "disable_first_eval", "output_dir", "gpus", "eval_steps", "num_epochs", "data_augment_noise",
                        "data_augment_rir", "train", "valid", "debug", "online", "no_freeze", "data_augment",
                        "batch_audio", "batch_audio_max_utts", "skip_unused_layers"))  # (the last three: this build's extensions, never part of the reference's name)
_N_DATA_OPTIONS = 8  # train, valid, debug, gpus, online, max_duration, min_duration, base_model: not learning hyper-parameters


def train_folder_name(options: Mapping[str, object], script_path: str, untrained: bool = False) -> str:
    """``hf_<md5 of (train, valid) paths>_<k>-<v>_..._adamwt[_nofreeze][_augment][_online]``; ``untrained`` keeps only the data
    options (the folder that caches the initial evaluation).  ``options`` in command-line declaration order."""
    items = list(options.items())
    if untrained:
        items = items[:_N_DATA_OPTIONS]
    name = "_".join(f"{_initials(k)}-{_compact(v)}" for k, v in items if k not in _NOT_NAMED)
    if not untrained:
        name += "_adamwt"
        for flag, tag in (("no_freeze", "_nofreeze"), ("data_augment", "_augment"), ("online", "_online")):
            if options.get(flag):
                name += tag
    if options.get("debug"):
        name = "DEBUG_" + name
    else:
        rel = strip_common_prefix([os.path.realpath(p) for p in (script_path, str(options["train"]), str(options["valid"]))], "/")
        name = hashmd5((rel[1], rel[2])) + "_" + name
    while "__" in name:
        name = name.replace("__", "_")
    return "hf_" + name


def _short_key(key: str, keep: int = 4) -> str:
    if len(key) <= keep:
        return key.capitalize()
    if "-" in key or "_" in key:
        return "".join(_short_key(part, 3) for part in key.replace("-", "_").split("_"))
    return key[:1].capitalize()


def hparams_to_str(options: Mapping[str, object], ignore: Iterable[str] = ("gpus", "gpu"), sort: bool = False) -> str:
    """Generic hyper-parameter string of ssak/utils/train_utils.py:4-38 (values that are existing dataset paths are not
    abbreviated here)."""
    fixed = {True: "1", False: "0", None: ""}
    items = sorted(options.items()) if sort else list(options.items())
    skip = set(ignore)
    text = "_".join(f"{_short_key(k)}-{fixed.get(v, str(v).replace('/', '_'))}" for k, v in items if k not in skip)
    while "__" in text:
        text = text.replace("__", "_")
    return text
