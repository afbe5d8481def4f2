"""Diagnostic: read a hipGraphDebugDotPrint file, report node/edge counts, longest path and any cycle."""
import re, sys
sys.setrecursionlimit(100000)
edges, nodes = {}, set()
for line in open(sys.argv[1]):
    m = re.match(r'\s*"?([^"\s\[]+)"?\s*->\s*"?([^"\s\[;]+)"?', line)
    if m:
        a, b = m.group(1), m.group(2)
        edges.setdefault(a, []).append(b); nodes.update((a, b))
print("nodes with edges", len(nodes), "edges", sum(len(v) for v in edges.values()))
self_loops = [(a, b) for a, bs in edges.items() for b in bs if a == b]
print("self loops", self_loops[:5])
dup = sum(len(v) - len(set(v)) for v in edges.values())
print("duplicate edges", dup)
color, cyc = {}, []
def dfs(u, path):
    color[u] = 1
    for v in edges.get(u, []):
        if color.get(v) == 1:
            cyc.append(path[path.index(v):] + [v] if v in path else [u, v]); return True
        if color.get(v) is None and dfs(v, path + [v]): return True
    color[u] = 2
    return False
for n in list(nodes):
    if color.get(n) is None and dfs(n, [n]): break
print("cycle", cyc[:1])
