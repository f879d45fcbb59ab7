"""Cheap static check that runs without a GPU: every global name a function of the host layer references exists
(the ops are only executed by the -m gpu tests, so a misspelt variable would otherwise surface on the GPU box only)."""
import builtins
import glob
import os
import symtable

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _walk(table, module_names, bad, path):
    for child in table.get_children():
        if child.get_type() == "function":
            for sym in child.get_symbols():
                if sym.is_global() and sym.is_referenced() and not sym.is_assigned():
                    n = sym.get_name()
                    if n not in module_names and not hasattr(builtins, n):
                        bad.append(f"{path}:{child.get_lineno()} {child.get_name()}() references undefined name {n!r}")
        _walk(child, module_names, bad, path)


def test_no_undefined_globals_in_host_layer():
    files = glob.glob(os.path.join(ROOT, "spatten_amd", "**", "*.py"), recursive=True) + \
        [os.path.join(ROOT, f) for f in ("bench.py", "__graft_entry__.py", "run_spatten_synthetic.py")] + \
        glob.glob(os.path.join(ROOT, "oracle", "*.py"))
    bad = []
    for f in files:
        src = open(f).read()
        top = symtable.symtable(src, f, "exec")
        module_names = {s.get_name() for s in top.get_symbols()} | {"__file__", "__name__", "__doc__"}
        _walk(top, module_names, bad, os.path.relpath(f, ROOT))
    assert not bad, "\n".join(bad)
