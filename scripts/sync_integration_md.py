#!/usr/bin/env python
"""INTEGRATION.md shows the code a maintainer adds to the reference; that code is COMPILED (integration/Makefile) and
tested (tests/test_gpu_integration_ref.py).  The markdown must therefore quote the compiled files, not a copy that can
drift: every

    <!-- snippet: integration/src/<file>#<tag> -->
    ```cpp
    ...
    ```

block is the text between `// [<tag>]` and `// [/<tag>]` of that file.  `--check` (what tests/test_integration_docs.py
runs) fails when the markdown differs; without it the markdown is rewritten in place."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARK = re.compile(r"<!-- snippet: (?P<path>[\w./-]+)#(?P<tag>[\w-]+) -->\n```cpp\n(?P<body>.*?)```\n", re.S)


def region(path, tag):
    text = open(os.path.join(ROOT, path)).read()
    m = re.search(r"// \[%s\]\n(.*?)// \[/%s\]" % (re.escape(tag), re.escape(tag)), text, re.S)
    if m is None:
        raise SystemExit("no region [%s] in %s" % (tag, path))
    return m.group(1)


def render(md):
    return MARK.sub(lambda m: "<!-- snippet: %s#%s -->\n```cpp\n%s```\n" % (m.group("path"), m.group("tag"),
                                                                              region(m.group("path"), m.group("tag"))), md)


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    md = open(path).read()
    new = render(md)
    n = len(MARK.findall(md))
    if "--check" in sys.argv:
        if new != md:
            raise SystemExit("INTEGRATION.md is out of sync with integration/src (run scripts/sync_integration_md.py)")
        print("INTEGRATION.md: %d snippet(s) in sync" % n)
        return
    open(path, "w").write(new)
    print("INTEGRATION.md: %d snippet(s) written" % n)


if __name__ == "__main__":
    main()
