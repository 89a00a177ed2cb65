"""The documents name files; the files exist. (Reference citations `pgsd.c:123` etc. are not repo
paths and are skipped.)"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", os.path.join("profiles", "README.md")]
PREFIXES = ("tests/", "tools/", "profiles/", "oracle/", "include/", "pgsd-sph_amd/", "bench.py", "__graft_entry__.py")


def candidates(text, doc):
    for m in re.finditer(r"`([^`\s]+)`", text):
        tok = m.group(1).split("::")[0]
        if "*" in tok or "<" in tok or "{" in tok or "|" in tok:
            continue
        if tok.startswith(PREFIXES):
            yield tok.rstrip(".,;:")
        elif doc.startswith("profiles") and re.match(r"^(r\d\d_|pack_traffic)[\w.]+\.(csv|json|jsonl|log|md)$", tok):
            yield "profiles/" + tok


def test_files_named_in_the_documents_exist():
    missing = []
    for doc in DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        for tok in candidates(text, doc):
            path = os.path.join(ROOT, tok)
            if tok.startswith("oracle/_ref") or tok.startswith("pgsd-sph_amd/csrc/build") or tok.startswith("tests/build") or tok.endswith(".so"):
                continue                      # built artefacts, not in the tree
            if not os.path.exists(path):
                missing.append((doc, tok))
    assert not missing, missing
