#!/usr/bin/env python3
"""INTEGRATION.md shows the reference-side binding; bindings/rust/*.rs ARE that binding (complete modules, uncompiled:
no Rust toolchain in this image).  Every fenced block that follows a marker line

    <!-- include: bindings/rust/NAME.rs -->

is replaced by the file's text, so the two cannot drift.  `--check` exits 1 if INTEGRATION.md is stale
(tests/test_host_logic.py runs it)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARK = re.compile(r"(<!-- include: (bindings/rust/[\w.]+) -->\n```rust\n)(.*?)(```\n)", re.S)


def render(text):
    def sub(m):
        with open(os.path.join(ROOT, m.group(2))) as f:
            body = f.read()
        if not body.endswith("\n"):
            body += "\n"
        return m.group(1) + body + m.group(4)
    return MARK.sub(sub, text)


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    with open(path) as f:
        old = f.read()
    new = render(old)
    included = set(m.group(2) for m in MARK.finditer(old))
    missing = [f"bindings/rust/{n}" for n in sorted(os.listdir(os.path.join(ROOT, "bindings", "rust")))
               if f"bindings/rust/{n}" not in included]
    if "--check" in sys.argv:
        if new != old or missing:
            print("INTEGRATION.md is stale (run tools/sync_integration.py)" if new != old else f"not shown in INTEGRATION.md: {missing}")
            return 1
        return 0
    with open(path, "w") as f:
        f.write(new)
    if missing:
        print("warning: not shown in INTEGRATION.md:", missing)
    return 0


if __name__ == "__main__":
    sys.exit(main())
