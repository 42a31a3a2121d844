"""Build stamps for everything under profiles/: which sources, which library, which commit an artefact was measured on.

    python tools/stamp.py write        (build container, before a gpurun call: records the commit -- the GPU box has no .git)
    python tools/stamp.py show
    python tools/stamp.py fetch r03    (build container, after the gpurun call: gpurun_out/r03_* -> profiles/, hashes checked)

``source_sha256`` = hash over the files the library is built from (ssak_amd/csrc/*, include/*, Makefile), ``lib_sha256`` = the
built libssak_hip.so.  tools/profile_round.sh embeds ``current()`` in every artefact it writes, bench.py prints it in its JSON
line and refuses HBM-traffic numbers measured on another library, tests/test_profiles.py checks that the artefacts of a round
agree with each other.
"""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library a run LOADS: SSAK_HIP_LIB redirects the binding (ssak_amd/hip.py), and an artefact must be labelled with what was measured
LIB = os.environ.get("SSAK_HIP_LIB") or os.path.join(ROOT, "ssak_amd", "lib", "libssak_hip.so")
COMMIT_FILE = os.path.join(ROOT, "profiles", ".commit.json")


def _sha(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.relpath(p, ROOT).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def source_sha256() -> str:
    files = sorted(glob.glob(os.path.join(ROOT, "ssak_amd", "csrc", "*")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    return _sha(files + [os.path.join(ROOT, "Makefile")])


def lib_sha256(path: str = LIB) -> str:
    return _sha([path]) if os.path.exists(path) else ""


def current() -> dict:
    """Stamp of the tree as it stands.  ``commit`` comes from git when there is one, else from profiles/.commit.json IF that was
    written for these very sources (else "unknown": a stale note must not label a newer tree)."""
    src = source_sha256()
    commit, dirty = "unknown", None
    try:
        commit = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "ssak_amd/csrc", "include", "Makefile"],
                                    capture_output=True, text=True, check=True).stdout.strip())
    except (OSError, subprocess.CalledProcessError):
        try:
            note = json.load(open(COMMIT_FILE))
            if note.get("source_sha256") == src:
                commit, dirty = note["commit"], note.get("dirty")
        except (OSError, ValueError, KeyError):
            pass
    st = {"commit": commit, "sources_modified_since_commit": dirty, "source_sha256": src, "lib_sha256": lib_sha256()}
    if os.environ.get("SSAK_HIP_LIB"):
        st["lib_path"] = os.path.relpath(LIB, ROOT)  # not the product library: an A/B or development build (its flags are not in source_sha256)
    return st


def fetch(round_name: str):
    """Build container, after a gpurun call ran tools/profile_round.sh: copy the round's artefacts from gpurun_out/ (what
    gpurun brings back) into profiles/, checking every file against the hashes the GPU box recorded."""
    import shutil
    src = os.path.join(ROOT, "gpurun_out")
    st = json.load(open(os.path.join(src, f"{round_name}_stamp.json")))
    for name, sha in st["files"].items():
        data = open(os.path.join(src, name), "rb").read()
        assert hashlib.sha256(data).hexdigest() == sha, f"gpurun_out/{name} is not the file the profile run wrote"
    for name in list(st["files"]) + [f"{round_name}_stamp.json"]:
        shutil.copyfile(os.path.join(src, name), os.path.join(ROOT, "profiles", name))
    print("profiles/: fetched", len(st["files"]) + 1, "files of", round_name, "measured on", st["stamp"]["commit"][:10],
          "library", st["stamp"]["lib_sha256"][:12])


def main():
    cmd = sys.argv[1] if len(sys.argv) > 1 else "show"
    if cmd == "fetch":
        return fetch(sys.argv[2])
    st = current()
    if cmd == "write":
        os.makedirs(os.path.dirname(COMMIT_FILE), exist_ok=True)
        json.dump({"commit": st["commit"], "dirty": st["sources_modified_since_commit"], "source_sha256": st["source_sha256"]},
                  open(COMMIT_FILE, "w"), indent=1)
    print(json.dumps(st))


if __name__ == "__main__":
    main()
