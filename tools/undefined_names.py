#!/usr/bin/env python3
"""Names a module's functions load but nothing defines (no pyflakes in the image): builtins, module-level bindings, imports (star
imports resolved by importing the module), function parameters and local stores are known; what is left is printed."""
import ast
import builtins
import importlib
import sys


def check(path):
    tree = ast.parse(open(path).read())
    known = set(dir(builtins))
    for node in ast.walk(tree):
        if isinstance(node, (ast.Import, ast.ImportFrom)):
            for a in node.names:
                if a.name == "*":
                    mod = importlib.import_module(("." * node.level) + node.module, package=path.rsplit("/", 1)[0].replace("/", ".") if node.level else None)
                    known.update(n for n in dir(mod) if not n.startswith("_"))
                else:
                    known.add((a.asname or a.name).split(".")[0])
        elif isinstance(node, (ast.FunctionDef, ast.ClassDef, ast.AsyncFunctionDef)):
            known.add(node.name)
            if not isinstance(node, ast.ClassDef):
                for a in node.args.args + node.args.kwonlyargs + node.args.posonlyargs + [x for x in (node.args.vararg, node.args.kwarg) if x]:
                    known.add(a.arg)
        elif isinstance(node, ast.Lambda):
            for a in node.args.args + node.args.kwonlyargs:
                known.add(a.arg)
        elif isinstance(node, ast.Name) and isinstance(node.ctx, (ast.Store, ast.Del)):
            known.add(node.id)
        elif isinstance(node, ast.ExceptHandler) and node.name:
            known.add(node.name)
    bad = sorted({(n.id, n.lineno) for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in known})
    for name, line in bad:
        print(f"{path}:{line}: undefined name {name}")
    return len(bad)


if __name__ == "__main__":
    sys.path.insert(0, ".")
    sys.exit(1 if sum(check(p) for p in sys.argv[1:]) else 0)
