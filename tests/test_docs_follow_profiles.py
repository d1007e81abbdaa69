"""The measured tables of DESIGN.md and README.md are GENERATED from the tracked files under profiles/ (tools/doc_tables.py):
this fails when a block in a document differs from what the script makes of the committed files -- a figure in such a
block therefore always exists in the profiles/ file the block names."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import doc_tables  # noqa: E402


def _blocks(text):
    return {m.group(1): m.group(2) for m in re.finditer(r"<!-- BEGIN GENERATED: (\w+) -->\n(.*?)<!-- END GENERATED: \1 -->", text, re.S)}


def test_generated_blocks_equal_the_profiles():
    want = doc_tables.render("r03")
    seen = set()
    for doc in ("DESIGN.md", "README.md"):
        found = _blocks(open(os.path.join(ROOT, doc)).read())
        for name, body in found.items():
            assert name in want, (doc, name)
            assert body == want[name] + "\n", f"{doc}: block {name} differs from tools/doc_tables.py's output: run `python tools/doc_tables.py --write`"
            seen.add(name)
    assert seen == set(want), sorted(set(want) - seen)  # every table the script knows is shown somewhere


def test_every_profile_file_a_block_names_is_tracked():
    text = open(os.path.join(ROOT, "DESIGN.md")).read() + open(os.path.join(ROOT, "README.md")).read()
    for block in _blocks(text).values():
        for name in re.findall(r"`profiles/([\w.<>-]+)`", block):
            if "<" in name:
                continue
            assert os.path.exists(os.path.join(ROOT, "profiles", name)), name
