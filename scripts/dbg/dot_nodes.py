"""Parse a verbose hipGraphDebugDotPrint dump: one line per node (id, kind, short label) with its predecessor ids."""
import re, sys
txt = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'"graph_0_node_(\d+)"\[[^\]]*?label="\{\s*(\w+)\s*\|(.*?)\}"\];', txt, re.S):
    nid, kind, body = int(m.group(1)), m.group(2), m.group(3)
    body = body.replace("\\<", "<").replace("\\>", ">").replace("\n", " ")
    if kind == "KERNEL":
        k = re.search(r"\{ID \| \d+ \| (\S+?)<<<\(([\d,]+)\)", body)
        name = k.group(1) if k else body[:80]
        name = re.sub(r"^_ZN?\d*", "", name)[:70] + " grid(" + (k.group(2) if k else "") + ")"
    else:
        name = re.sub(r"\s+", " ", body)[:200]
    nodes[nid] = (kind, name)
preds = {}
for m in re.finditer(r'"graph_0_node_(\d+)"\s*->\s*"graph_0_node_(\d+)"', txt):
    preds.setdefault(int(m.group(2)), []).append(int(m.group(1)))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = int(sys.argv[3]) if len(sys.argv) > 3 else 10**9
for nid in sorted(nodes):
    if lo <= nid <= hi:
        print(nid, nodes[nid][0], nodes[nid][1], "<-", sorted(preds.get(nid, [])))
