"""Scope-aware undefined-name scan (a small stand-in for pyflakes, which this image does not ship): every name loaded in a
function / lambda / class body must be bound in that scope, an enclosing function scope, the module, or builtins.  Exists
because most of the engine and half of the tests only execute on a GPU box: a NameError there costs GPU minutes.
Usage: python tests/tools/undefined_names.py FILES...   (prints findings; tests/test_static_names.py asserts none)"""
import ast, builtins, sys
B = set(dir(builtins)) | {"__file__", "__name__", "__doc__", "__class__"}

def local_bindings(fn):
    """names bound directly in this scope (not in nested function/class bodies)"""
    names = set()
    def visit(n, top=True):
        for c in ast.iter_child_nodes(n):
            if isinstance(c, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                names.add(c.name)
                # decorators / defaults evaluated in this scope but bind nothing
                continue
            if isinstance(c, ast.Lambda):
                continue
            if isinstance(c, ast.Name) and isinstance(c.ctx, (ast.Store, ast.Del)):
                names.add(c.id)
            elif isinstance(c, (ast.Import, ast.ImportFrom)):
                for a in c.names: names.add((a.asname or a.name).split(".")[0])
            elif isinstance(c, ast.ExceptHandler) and c.name:
                names.add(c.name)
            elif isinstance(c, (ast.Global, ast.Nonlocal)):
                names.update(c.names)
            elif isinstance(c, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
                # comprehension targets are local to the comprehension; treat as visible inside it only -> handled in loads()
                pass
            visit(c, False)
    visit(fn)
    if isinstance(fn, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
        a = fn.args
        for x in a.posonlyargs + a.args + a.kwonlyargs: names.add(x.arg)
        if a.vararg: names.add(a.vararg.arg)
        if a.kwarg: names.add(a.kwarg.arg)
    return names

def comp_targets(n):
    t = set()
    for g in n.generators:
        for x in ast.walk(g.target):
            if isinstance(x, ast.Name): t.add(x.id)
    return t

def check_scope(node, visible, path, is_class=False):
    mine = local_bindings(node)
    vis = visible | mine
    def walk(n, extra):
        for c in ast.iter_child_nodes(n):
            if isinstance(c, (ast.FunctionDef, ast.AsyncFunctionDef)):
                for d in c.decorator_list + c.args.defaults + [x for x in c.args.kw_defaults if x]:
                    walk_expr(d, extra)
                # class-level names are NOT visible inside methods
                check_scope(c, (visible if is_class else vis) | extra, path)
            elif isinstance(c, ast.ClassDef):
                for d in c.decorator_list + c.bases: walk_expr(d, extra)
                check_scope(c, (visible if is_class else vis) | extra, path, is_class=True)
            elif isinstance(c, ast.Lambda):
                check_scope(c, vis | extra, path)
            elif isinstance(c, (ast.ListComp, ast.SetComp, ast.DictComp, ast.GeneratorExp)):
                walk(c, extra | comp_targets(c))
            elif isinstance(c, ast.Name):
                if isinstance(c.ctx, ast.Load) and c.id not in vis and c.id not in extra and c.id not in B:
                    print(f"{path}:{c.lineno}: undefined name '{c.id}'")
            else:
                walk(c, extra)
    def walk_expr(e, extra):
        class W: pass
        m = ast.Module(body=[ast.Expr(e)], type_ignores=[])
        walk(m, extra)
    walk(node, set())

def scan(paths):
    import contextlib, io
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        for p in paths:
            try:
                tree = ast.parse(open(p).read())
            except SyntaxError as e:
                print(p, "SYNTAX", e); continue
            check_scope(tree, set(), p)
    return [l for l in buf.getvalue().splitlines() if l]


if __name__ == "__main__":
    for line in scan(sys.argv[1:]):
        print(line)
