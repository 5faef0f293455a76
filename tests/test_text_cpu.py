"""CPU: the text front end against token ids produced by the reference's own TextCleaner (tests/golden/text_golden.json)."""
import json
import os

from artspeech_amd.text import TextCleaner, symbols


def test_symbol_table_and_ids(golden_dir):
    with open(os.path.join(golden_dir, "text_golden.json"), encoding="utf-8") as f:
        g = json.load(f)
    assert symbols == g["symbols"] and len(symbols) == 178
    tc = TextCleaner()
    assert g["cases"]
    for case in g["cases"]:
        assert tc(case["text"]) == case["ids"]
    assert tc("$") == [0]
    assert tc("a中b") == tc("ab")       # characters outside the table are dropped (test.py:36-37)
