#!/usr/bin/env python3
"""Build-container tool: a fixture for trajectory_io.py from the one trajectory file the reference ships,
src/routes.h — written by the reference's own GUI (gui/gui_manager.py:442-507), i.e. reference OUTPUT.

    python oracle/gen_routes_fixture.py            # writes tests/golden/files/routes_h_slice.json

The route in that file has ~650 rows; the fixture keeps the file's skeleton lines verbatim, the route's name and
the exact text of a slice of its rows (the first 48, every action row with the row after it, the last 6), so the test
can hold routes_header_entry / update_routes_header / format_txt to the reference's number formatting and separators
without the repository carrying the whole 81 KB file.  Data only: no reference code is read or copied.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = "/root/reference/src/routes.h"


def main():
    text = open(SRC).read()
    lines = text.split("\n")
    entry_idx = [i for i, l in enumerate(lines) if l.startswith("std::vector<std::vector<double>>")]
    assert len(entry_idx) == 1, "one route expected"
    entry = lines[entry_idx[0]]
    m = re.match(r"std::vector<std::vector<double>> (\w+) = \{(.*)\};$", entry)
    name, body = m.group(1), m.group(2)
    rows = re.findall(r"\{([^{}]*)\}", body)
    assert "{" + "}, {".join(rows) + "}" == body, "row split does not reproduce the line"
    keep = set(range(48)) | set(range(len(rows) - 6, len(rows)))
    for i, r in enumerate(rows):
        if r.split(", ")[0] == "1":
            keep |= {i, min(i + 1, len(rows) - 1)}
    keep = sorted(keep)
    out = {
        "source": "src/routes.h of the reference (output of gui/gui_manager.py:442-507)",
        "name": name,
        "n_rows_in_file": len(rows),
        "kept_rows": keep,
        "rows_text": [rows[i] for i in keep],
        "skeleton": [l if i != entry_idx[0] else "<ENTRY>" for i, l in enumerate(lines)],
    }
    dst = os.path.join(ROOT, "tests", "golden", "files", "routes_h_slice.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", dst, len(keep), "of", len(rows), "rows")


if __name__ == "__main__":
    sys.exit(main())
