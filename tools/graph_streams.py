"""Which executor stream does every node of a captured step run on?  Parses the DOT file ROCm writes at hipGraphInstantiate under
DEBUG_HIP_GRAPH_DOT_PRINT=1 (every node is labelled with its StreamId and whether a cross-stream signal is required) and reports

  * nodes per stream, cross-stream edges, nodes with signals;
  * every pair of INDEPENDENT chains (neither can reach the other) that the executor put on ONE stream: such a pair runs one
    after the other whatever HIP streams it was captured on - the schedule of DESIGN 2.9 is built to keep this list empty.

    DEBUG_HIP_GRAPH_DOT_PRINT=1 python tools/graph_streams.py --capture [model]      # writes graph_<pid>_dot_print_* into the cwd, then analyses
    python tools/graph_streams.py graph_1234_dot_print_1"""
import collections
import glob
import os
import re
import subprocess
import sys


def parse(path):
    txt = open(path).read()
    nodes = {}
    for m in re.finditer(r'"graph_\d+_node_(\d+)"\[[^\]]*?label="\d+\n([^\n]*)\n(?:\([^\n]*\)\n)?StreamId:(\d+)\nSignalIsRequired: (\w+)', txt):
        nodes[int(m.group(1))] = (m.group(2), int(m.group(3)), m.group(4) == 'true')
    edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_\d+_node_(\d+)"\s*->\s*"graph_\d+_node_(\d+)"', txt)]
    return nodes, edges


def short(n):
    try:
        d = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip()
    except Exception:
        d = n
    d = d.replace('(anonymous namespace)::', '').replace('void ', '')
    return re.sub(r'\(.*', '', d)[:44]


def report(path):
    nodes, edges = parse(path)
    succ, pred = collections.defaultdict(list), collections.defaultdict(list)
    for a, b in edges:
        succ[a].append(b)
        pred[b].append(a)
    per = collections.Counter(v[1] for v in nodes.values())
    cross = sum(1 for a, b in edges if nodes[a][1] != nodes[b][1])
    print(f'{path}: {len(nodes)} nodes, {len(edges)} edges ({cross} cross-stream), {sum(v[2] for v in nodes.values())} nodes with signals; '
          f'nodes per stream {dict(sorted(per.items()))}')
    import functools
    import sys as _sys
    _sys.setrecursionlimit(100000)

    @functools.lru_cache(maxsize=None)
    def reach(a):
        out = set()
        stack = list(succ[a])
        while stack:
            v = stack.pop()
            if v in out:
                continue
            out.add(v)
            stack.extend(succ[v])
        return frozenset(out)

    def chain_len(c):                             # nodes that follow c on its own stream before the stream changes hands
        n, cur = 1, c
        while True:
            nxt = [v for v in succ[cur] if nodes[v][1] == nodes[c][1] and len(pred[v]) == 1]
            if len(nxt) != 1:
                return n
            cur = nxt[0]
            n += 1
    # chains = maximal runs of nodes on one stream linked by single-predecessor edges; two chains of ONE stream that cannot reach
    # each other are independent work the executor runs one after the other (in enqueue order)
    head_of, chains = {}, {}
    for n in sorted(nodes):
        p = [q for q in pred[n] if nodes[q][1] == nodes[n][1]]
        if len(pred[n]) == 1 and len(p) == 1 and len([v for v in succ[p[0]] if nodes[v][1] == nodes[n][1] and len(pred[v]) == 1]) == 1:
            head_of[n] = head_of.get(p[0], p[0])
        else:
            head_of[n] = n
        chains.setdefault(head_of[n], []).append(n)
    long = {h: c for h, c in chains.items() if len(c) >= 4}
    alias = []
    heads = sorted(long)
    for i, a in enumerate(heads):
        for b in heads[i + 1:]:
            if nodes[a][1] != nodes[b][1]:
                continue
            ta, tb = long[a][-1], long[b][-1]
            if b in reach(ta) or a in reach(tb) or b in reach(a) or a in reach(b):
                continue
            alias.append((a, b))
    for a, b in alias[:12]:
        print(f'  stream {nodes[a][1]}: independent chains {a}:{short(nodes[a][0])[:26]} ({len(long[a])} nodes) and '
              f'{b}:{short(nodes[b][0])[:26]} ({len(long[b])} nodes)')
    print(f'{len(alias)} pairs of INDEPENDENT chains (>= 4 nodes each) share a stream: each such pair runs one after the other')


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == '--capture':
        ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, DEBUG_HIP_GRAPH_DOT_PRINT='1')
        model = sys.argv[2] if len(sys.argv) > 2 else 't_nus_bn'
        subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--model', model, '--steps', '2', '--warmup', '1', '--no-cpu-baseline',
                        '--no-neck', '--no-eager', '--no-roofline'], env=env, stdout=subprocess.DEVNULL)
        files = sorted(glob.glob('graph_*_dot_print_*'), key=os.path.getsize)
        if not files:
            raise SystemExit('no DOT file written')
        report(files[-1])                       # the training step is the largest graph of the run
    else:
        for p in sys.argv[1:]:
            report(p)
