"""Every file the documents cite exists and is not empty.

Round 4 shipped two empty overlap profiles that README / DESIGN quoted as evidence.  The trace tools
have their own test since (tests/test_tools_cpu.py); this one covers the other half: a path written between backticks in
DESIGN.md, DESIGN_HISTORY.md, README.md, INTEGRATION.md or profiles/README.md - a profile, a test, a tool, a source file - must be there.
Built artefacts (git-ignored binaries the documents name beside their sources) are the only exceptions, and only when the
source they are built from exists."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ("DESIGN.md", "DESIGN_HISTORY.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md"))
TOKEN = re.compile(r"`([A-Za-z0-9_./\-]+)`")
DIRS = ("profiles/", "tests/", "tools/", "oracle/", "include/", "examples/", "ek-pnp-3d_amd/", "csrc/")
PROFILE_NAME = re.compile(r"r0\d[a-z]?_[A-Za-z0-9_.\-]+\.(json|jsonl|csv|log|txt)$")


def _cited(doc):
    for m in TOKEN.finditer(open(os.path.join(ROOT, doc), encoding="utf-8").read()):
        t = m.group(1).rstrip(".")
        if t.startswith("csrc/"):
            yield t, os.path.join("ek-pnp-3d_amd", t)
        elif t.startswith(DIRS):
            yield t, t
        elif PROFILE_NAME.match(t) or t == "pmc_traffic.json":
            yield t, os.path.join("profiles", t)


def _built_from_existing_source(rel):
    if rel.startswith("oracle/_ref"):  # the reference compiled here by oracle/build_ref.sh (git-ignored, travels to the GPU box)
        return os.path.exists(os.path.join(ROOT, "oracle", "build_ref.sh"))
    if rel.startswith("tools/") and "." not in os.path.basename(rel):  # a probe binary beside its .hip source
        return os.path.exists(os.path.join(ROOT, rel + ".hip"))
    if rel in ("ek-pnp-3d_amd/libekpnp.so", "ek-pnp-3d_amd/ekpnp_main", "oracle/libekpnp_oracle.so"):
        return True
    return False


def test_every_cited_file_exists_and_is_not_empty():
    bad, seen = [], 0
    for doc in DOCS:
        for shown, rel in _cited(doc):
            seen += 1
            p = os.path.join(ROOT, rel)
            if os.path.isdir(p) or (os.path.isfile(p) and os.path.getsize(p) > 0):
                continue
            if not os.path.exists(p) and _built_from_existing_source(rel):
                continue
            bad.append(f"{doc}: `{shown}`")
    assert seen > 200  # the pattern still finds the citations
    assert not bad, "cited but missing or empty:\n  " + "\n  ".join(sorted(set(bad)))


def test_json_profiles_parse_and_say_something():
    """every .json / .jsonl under profiles/ is valid JSON and not an empty container"""
    import json

    prof = os.path.join(ROOT, "profiles")
    n = 0
    for f in sorted(os.listdir(prof)):
        p = os.path.join(prof, f)
        if f.endswith(".json"):
            d = json.load(open(p))
            assert d not in ({}, [], None), f
            if isinstance(d, dict) and "steps_seen" in d:
                assert d["steps_seen"] > 0 and d.get("steps"), f  # (round 4's empty overlap files looked like this)
            n += 1
        elif f.endswith(".jsonl"):
            lines = [ln for ln in open(p).read().splitlines() if ln.strip().startswith("{")]
            assert lines, f
            for ln in lines:
                json.loads(ln)
            n += 1
    assert n > 50
