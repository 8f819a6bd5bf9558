"""Maintain profiles/pmc_traffic.json: per-kernel HBM traffic / MFMA-busy figures from rocprofv3 PMC passes, each entry tied to
the sha256 of the source files that hold the kernel at the time it was measured.  bench.py prints an entry's `traffic` only
while those files are unchanged (bench.pmc_entry), so the JSON line never carries a counter value of a kernel that has since
been edited.

    python profiles/src/pmc_traffic.py show
    python profiles/src/pmc_traffic.py set NAME --files video_r2_amd/csrc/gemv.hip video_r2_amd/csrc/common.h \
        --source profiles/r04_pmc_x.md traffic_bytes=285.3e6 fetch_size_kib=136939 write_size_kib=4736 [mfma_busy_pct=63.5]
    python profiles/src/pmc_traffic.py set NAME --at-commit f314d06 ...     (hash the files as they were at that commit)
"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PATH = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def sha(path, commit=None):
    if commit:
        data = subprocess.run(["git", "-C", ROOT, "show", f"{commit}:{path}"], check=True, capture_output=True).stdout
    else:
        with open(os.path.join(ROOT, path), "rb") as f:
            data = f.read()
    return hashlib.sha256(data).hexdigest()


def load():
    if os.path.exists(PATH):
        with open(PATH) as f:
            return json.load(f)
    return {"_about": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction of MI355X_MICROARCH.md) and MFMA-busy "
                      "share from separate rocprofv3 --pmc passes; an entry is valid while the sha256 of its files is unchanged",
            "entries": {}}


def main(argv):
    db = load()
    if not argv or argv[0] == "show":
        for name, e in db["entries"].items():
            ok = all(sha(p) == h for p, h in e["files"].items())
            print(("valid  " if ok else "STALE  ") + name, {k: v for k, v in e.items() if k != "files"})
        return
    if argv[0] != "set":
        raise SystemExit(__doc__)
    name, rest = argv[1], argv[2:]
    files, source, commit, vals = [], None, None, {}
    i = 0
    while i < len(rest):
        if rest[i] == "--files":
            i += 1
            while i < len(rest) and not rest[i].startswith("--") and "=" not in rest[i]:
                files.append(rest[i]); i += 1
            continue
        if rest[i] == "--source":
            source = rest[i + 1]; i += 2; continue
        if rest[i] == "--at-commit":
            commit = rest[i + 1]; i += 2; continue
        k, v = rest[i].split("=", 1)
        try:
            vals[k] = float(v)
        except ValueError:
            vals[k] = v
        i += 1
    e = dict(vals)
    e["source"] = source
    e["files"] = {p: sha(p, commit) for p in files}
    if commit:
        e["measured_at_commit"] = commit
    db["entries"][name] = e
    with open(PATH, "w") as f:
        json.dump(db, f, indent=1, sort_keys=True)
        f.write("\n")


if __name__ == "__main__":
    main(sys.argv[1:])
