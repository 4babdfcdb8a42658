"""Minimal pytrees: tuples, lists, dicts, None; everything else is a leaf (refshim)."""


def tree_leaves(tree):
    if tree is None:
        return []
    if isinstance(tree, (tuple, list)):
        return [leaf for t in tree for leaf in tree_leaves(t)]
    if isinstance(tree, dict):
        return [leaf for k in sorted(tree) for leaf in tree_leaves(tree[k])]
    return [tree]


def tree_map(f, tree, *rest):
    if tree is None:
        return None
    if isinstance(tree, (tuple, list)):
        return type(tree)(tree_map(f, t, *[r[i] for r in rest]) for i, t in enumerate(tree))
    if isinstance(tree, dict):
        return {k: tree_map(f, v, *[r[k] for r in rest]) for k, v in tree.items()}
    return f(tree, *rest)


def tree_reduce(f, tree, *init):
    import functools

    return functools.reduce(f, tree_leaves(tree), *init)
