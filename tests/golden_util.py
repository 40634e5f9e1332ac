"""Column alignment for golden comparisons: the reference orders its het_* columns by iterating a Python set
(popgenWindows.py:277-280, 347), i.e. by string hash; the same columns in another order are the same result."""


def align_columns(got, want):
    g, w = got.splitlines(), want.splitlines()
    if not g or not w or g[0] == w[0]:
        return got
    gh, wh = g[0].split(","), w[0].split(",")
    if len(gh) != len(wh) or sorted(gh) != sorted(wh) or len(set(gh)) != len(gh):
        return got
    perm = [gh.index(name) for name in wh]
    out = []
    for line in g:
        cells = line.split(",")
        out.append(",".join(cells[k] for k in perm) if len(cells) == len(gh) else line)
    return "\n".join(out) + ("\n" if got.endswith("\n") else "")
