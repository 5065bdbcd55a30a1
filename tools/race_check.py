"""Cross-lane race check: the gradients of ONE eager training step must not depend on the relative timing of the lanes.
The same step (same weights, inputs, cotangents; stochastic layers off) is run under timing perturbations - one leaf lane
instead of four, idle launches in front of the blocks of one width (HRF_DEBUG_PAD), no grouping - and every gradient tensor is
compared with the first run: rel-L2 <= 1e-3 passes (two UNPERTURBED runs differ by up to 5e-4 in a few BatchNorm-gamma gradients
that are small differences of large sums - the order of the fp32 atomics upstream; everything else repeats to 1e-5).  A writer that overwrites a buffer
another lane accumulates into shows up as a per-tensor difference of several percent (found this way: the two transition1
convolutions of the plain HRFormer).  For the HRFuser models the captured step (hipGraph replay: another stream assignment) is replayed 8 times
against the eager step as well.   python tools/race_check.py [t_nus_bn|b_nus_bn|t_stf_bn|hrformer_t_bn|hrnet|stage_d ...]"""
import copy, json, os, sys
os.environ['HRF_MODULE_GRAPH'] = '0'              # eager launches on the lanes: the timing perturbations must act on every call
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import helpers as T                                 # noqa: E402
from hrfuser_amd import build_backbone              # noqa: E402

PERTURB = [{}, {'HRF_WGRAD_LANES': '1'}, {'HRF_DEBUG_PAD': '18:40'}, {'HRF_DEBUG_PAD': '36:40'}, {'HRF_DEBUG_PAD': '72:40,144:40'},
           {'HRF_WGRAD_GROUP': '0'}, {'HRF_DEBUG_PAD': '78:40,156:40'}, {}]


def check(tag):
    dev = torch.device('cuda:0')
    gold = os.path.join(ROOT, 'tests', 'golden')
    if tag.startswith('hrformer'):
        cfg, mc, size = json.load(open(os.path.join(gold, 'hrformer_cfgs.json')))[tag], [], (2, 64, 96)
    else:
        if tag == 'hrnet':
            cfg = json.load(open(os.path.join(gold, 'hrfuser_hrnet_cfg.json')))['cfg']
        elif tag == 'stage_d':
            cfg = json.load(open(os.path.join(gold, 'backbone_cfg_stage_d.json')))['t_nus_bn_stage_d']
        else:
            cfg = json.load(open(os.path.join(gold, 'backbone_cfgs.json')))[tag]
        mc, size = cfg.get('mod_in_channels', [3, 3]), (2, 64, 96)
    torch.manual_seed(0)
    net = build_backbone(copy.deepcopy(cfg)).to(dev)
    T.disable_stochastic(net)
    net.train(True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(size[0], 3, size[1], size[2], generator=g)
    mods = [torch.randn(size[0], c, size[1], size[2], generator=g) for c in mc]

    def run(env):
        for k, v in env.items():
            os.environ[k] = v
        for p in net.parameters():
            if p.grad is not None:
                p.grad.zero_()
        xa = x.clone().to(dev).requires_grad_(True)
        ms = [m.clone().to(dev).requires_grad_(True) for m in mods]
        ya = net(xa, ms) if mods else net(xa)
        gg = torch.Generator().manual_seed(5)
        cots = [torch.randn(t.shape, generator=gg) for t in ya]
        sum((t * c.to(dev)).sum() for t, c in zip(ya, cots)).backward()
        torch.cuda.synchronize()
        out = {'__x': xa.grad.detach().cpu().clone()}
        out.update({f'__mod{k}': m.grad.detach().cpu().clone() for k, m in enumerate(ms)})
        out.update({n: p.grad.detach().cpu().clone() for n, p in net.named_parameters() if p.grad is not None})
        for k in env:
            os.environ.pop(k, None)
        return out
    ref = run({})
    nmax = max(float(v.double().norm()) for v in ref.values())
    worst_all = 0.0
    for env in PERTURB:
        cur = run(env)
        bad, worst = [], 0.0
        for k, a in ref.items():
            a, b = a.double(), cur[k].double()
            if float(a.norm()) < 1e-6 * nmax:
                continue
            e = float((a - b).norm() / float(a.norm()))
            worst = max(worst, e)
            if e > 1e-3:
                bad.append((k, round(e, 5)))
        worst_all = max(worst_all, worst)
        print(f'{tag:14s} {str(env):44s} worst rel-L2 {worst:.2e}  tensors above 1e-3: {len(bad)} {bad[:6]}')
    return worst_all


def check_captured(tag, size=(2, 192, 320), replays=8):
    """The product path: ONE captured training step (hrfuser_amd.trainer.Trainer, learning rate 0 so that the parameters stay put)
    replayed `replays` times must reproduce the gradient arena of the eager step (rel-L2 of the whole arena)."""
    from hrfuser_amd.trainer import Trainer, make_cotangents
    dev = torch.device('cuda:0')
    cfg = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'backbone_cfgs.json')))[tag]
    mc = cfg.get('mod_in_channels', [3, 3])
    torch.manual_seed(0)
    net = build_backbone(copy.deepcopy(cfg)).to(dev)
    T.disable_stochastic(net)
    net.train(True)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(size[0], 3, size[1], size[2], generator=g).to(dev)
    mods = [torch.randn(size[0], c, size[1], size[2], generator=g).to(dev) for c in mc]
    cots = make_cotangents(net, x, mods)
    tr = Trainer(net, lr=0.0)
    eng = net._engine()
    tr.step(x, mods, cots)
    torch.cuda.synchronize()
    ref = eng.flat_g.detach().clone()
    tr.capture(x, mods, cots)
    worst = 0.0
    for _ in range(replays):
        tr.replay()
        torch.cuda.synchronize()
        worst = max(worst, float((eng.flat_g.detach() - ref).norm() / ref.norm()))
    print(f'{tag:14s} {replays} captured replays vs the eager step: worst rel-L2 of the gradient arena {worst:.2e}')
    return worst


def check_two_process_p2p():
    """Two ranks sharing this GPU through the peer-to-peer SyncBN exchange, lane timing perturbed on ONE of them
    (tests/test_p2p_exchange.py, mode 'skew'): -> worst rel-L2 against the unperturbed step."""
    import re
    import test_p2p_exchange as TP
    outs = TP._run_two('skew', 29681, timeout=600)
    worst = 0.0
    for rc, o, e in outs:
        m = re.findall(r'"rel_l2_vs_unperturbed": ([0-9.e+-]+)', o)
        if 'P2P_SKEW_OK' not in o or not m:
            print('two-process peer-to-peer arm FAILED', rc, o[-400:], e[-800:])
            return 1.0
        worst = max(worst, float(m[-1]))
    print(f'two processes, peer-to-peer exchange, one rank perturbed: worst rel-L2 {worst:.2e}')
    return worst


if __name__ == '__main__':
    tags = sys.argv[1:] or ['t_nus_bn', 'b_nus_bn', 't_stf_bn', 'hrformer_t_bn', 'hrnet', 'stage_d']
    w = max(check(t) for t in tags)
    w = max([w] + [check_captured(t) for t in tags if t in ('t_nus_bn', 'b_nus_bn', 't_stf_bn')])
    if not sys.argv[1:]:
        w = max(w, check_two_process_p2p())
    print('RACE CHECK', 'OK' if w <= 1e-3 else 'FAILED', f'(worst {w:.2e})')
    sys.exit(0 if w <= 1e-3 else 1)
