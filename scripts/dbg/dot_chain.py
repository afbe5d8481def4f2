"""Print, for a hipGraph DOT dump, every node with its label and predecessors, in file order (to read dependency chains)."""
import re, sys
labels, preds, order = {}, {}, []
for line in open(sys.argv[1]):
    m = re.match(r'\s*"?([^"\s\[]+)"?\s*->\s*"?([^"\s\[;]+)"?', line)
    if m:
        preds.setdefault(m.group(2), []).append(m.group(1))
        continue
    m = re.match(r'\s*"?([^"\s\[]+)"?\s*\[(.*)\]', line)
    if m:
        lab = re.search(r'label="([^"]*)"', m.group(2))
        labels[m.group(1)] = (lab.group(1) if lab else m.group(2))[:110].replace("\\n", " | ")
        order.append(m.group(1))
pat = sys.argv[2] if len(sys.argv) > 2 else None
for i, n in enumerate(order):
    if pat and not re.search(pat, labels[n]):
        continue
    print(i, n, labels[n], "<-", [f"{order.index(p) if p in order else '?'}" for p in preds.get(n, [])])
