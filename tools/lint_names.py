#!/usr/bin/env python3
"""Undefined-name check without a linter (the image has none, and host-side Python errors otherwise only surface on the GPU box):
every name a scope reads as a global must be defined at module level or be a builtin.   usage: lint_names.py file.py ..."""
import builtins
import symtable
import sys


def walk(tab, module_names, path, out):
    for s in tab.get_symbols():
        if s.is_referenced() and s.is_global() and not s.is_assigned() and s.get_name() not in module_names and not hasattr(builtins, s.get_name()):
            out.append(f"{path}:{tab.get_lineno()}: `{s.get_name()}` read in {tab.get_type()} `{tab.get_name()}` is not defined at module level")
    for ch in tab.get_children():
        walk(ch, module_names, path, out)


def main(paths):
    out = []
    for p in paths:
        src = open(p).read()
        top = symtable.symtable(src, p, "exec")
        names = {s.get_name() for s in top.get_symbols() if s.is_assigned() or s.is_imported() or s.is_namespace()}
        names |= {"__file__", "__name__", "__doc__"}
        walk(top, names, p, out)
    print("\n".join(out) if out else "ok")
    return 1 if out else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
