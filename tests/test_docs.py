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

DOCS = ["DESIGN.md", "HISTORY.md", "BASELINE.md", "README.md", "INTEGRATION.md", "profiles/round3/README.md", "profiles/round4/README.md", "profiles/round5/README.md", "profiles/round6/README.md",
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
    for ledger in ("## Round 6", "## Round 5", "## Round 4", "## Round 3 ledger", "## Round 2 ledger"):
        assert ledger in history and ledger not in design
    # the lines the round-3 review named as stale are gone
    assert "skewed to 17 mod 32" not in design and "skewed to 17 mod 32 floats.  512-point" not in history
    assert "512 MiB 23.0" in history and "26.6 ms" in history             # all four ring-rotate rows, group = 32 beside them
    # README's headline is the driver's figure: the NEWEST BENCH_rNN.json in the tree, quoted to one decimal
    # (the newest one git TRACKS: the driver writes this round's BENCH file after the last commit of the round, and no
    # document of the tree can quote a figure that does not exist yet)
    benches = sorted(os.path.join(ROOT, f) for f in _tracked() if re.fullmatch(r"BENCH_r\d+\.json", f))
    assert benches
    newest = benches[-1]
    import json
    parsed = json.load(open(newest))["parsed"]
    readme = open(os.path.join(ROOT, "README.md")).read()
    assert os.path.basename(newest) in readme and os.path.basename(newest) in design
    assert f"{parsed['value']:.1f}" in readme and f"{parsed['roofline']['frac']:.3f}" in readme, (newest, parsed["value"])
    assert f"{parsed['value']:.1f}" in design or f"{parsed['value']:.2f}" in design


def _numbers(text):
    return [t for t in re.findall(r"(?<![\w.^])\d+(?:\.\d+)?(?![\w^])", text)]


def test_profile_index_rows_quote_numbers_that_are_in_their_files():
    """every number a `profiles/roundN/README.md` row quotes for its `a_*` files occurs in one of
    those files (round 5's index described an overwritten set).  A quoted number matches a number of the file when the file's
    number, rounded to the quoted precision, equals it -- also across a unit step of 1000 (GB/s quoted as TB/s, us as ms).
    Figures inside parentheses are derived ones and are not checked; integers below 100 (counts, exponents) neither."""
    checked = 0
    for rnd in (5, 6):
        text = open(os.path.join(ROOT, "profiles", f"round{rnd}", "README.md")).read()
        for line in text.split("\n"):
            cells = [c.strip() for c in line.split("|")]
            if len(cells) < 4 or not cells[1].startswith("`a_"):
                continue
            files = re.findall(r"`([A-Za-z0-9_.]+)`", cells[1])
            assert files and all(f.startswith("a_") for f in files), line[:80]
            have = []
            for f in files:
                raw = open(os.path.join(ROOT, "profiles", f"round{rnd}", f)).read()
                have += [float(t) for t in re.findall(r"(?<![\w.])-?\d+(?:\.\d+)?(?:[eE][-+]?\d+)?", raw)]
            claim = re.sub(r"\([^()]*\)", "", cells[2])          # derived figures live in parentheses
            claim = re.sub(r"`[^`]*`", "", claim)                 # names, flags and keys in backticks are not figures
            for tok in _numbers(claim):
                if "." not in tok and int(tok) < 100:
                    continue
                dec = len(tok.split(".")[1]) if "." in tok else 0
                q = float(tok)
                ok = any(round(h * k, dec) == q for h in have for k in (1.0, 1e-3, 1e3))
                assert ok, (rnd, files, tok)
                checked += 1
    assert checked >= 25, checked


def test_bench_reference_belongs_to_the_kernel_sources_in_the_tree():
    """`roofline.traffic` of the bench line is a stored PMC figure (profiles/bench_reference.json).  It is
    stamped with the commit and the sha256 of the two kernel sources it was profiled on; a change to those sources without a new
    profile job (tools/run_round_profiles.sh) fails HERE, before bench.py has to say `"traffic_stale": true` in the driver's line."""
    import hashlib
    import json
    ref = json.load(open(os.path.join(ROOT, "profiles", "bench_reference.json")))
    stamp = ref["taken_on"]
    assert stamp["kernel_sources"] == ["fft_wgpu_amd/csrc/tile_1m.h", "fft_wgpu_amd/csrc/kernels_1m.hip"]
    h = hashlib.sha256()
    for rel in stamp["kernel_sources"]:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    assert stamp["kernel_source_sha256"] == h.hexdigest(), "kernel sources changed since the PMC passes: re-run tools/run_round_profiles.sh"
    assert re.fullmatch(r"[0-9a-f]{7,40}", stamp["commit"]) and stamp["round"] >= 6
    assert f"profiles/round{stamp['round']}/a_pmc_fetch_summary.txt" in ref["traffic_source"]
    # bench.py hashes the same list and reports the same stamp
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert '"traffic_stale": traffic_stale' in bench and '"traffic_taken_on": taken_on' in bench
