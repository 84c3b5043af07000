"""EV2H_FPS_CHUNKS: chunked enc.sa1 sampling against the one-launch schedule -- bit-identical outputs, and what it buys at the small
shapes:  python tools/debug/chunk_check.py"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
from ev2hands_amd import synth
from ev2hands_amd.model import TEHNetWrapper
os.environ["ERPC"] = "0"
net = TEHNetWrapper("cuda:0", mano_assets={s: synth.synth_mano_assets(s, 0) for s in ("left", "right")}, precision=sys.argv[1])
net.load_state_dict(synth.synth_state_dict(4, 0), strict=True); net.eval()
out = {}
for B, N in eval(os.environ.get("CHUNK_SHAPES", "((16, 8192), (8, 2048), (1, 2048), (32, 2048))")):
    x = synth.synth_cloud("E", B, 4, N, 5).cuda(); inits = synth.fps_inits(B, N, 5)
    def once():
        net.net.fps_init = inits
        with torch.no_grad():
            return net(x)
    o = once(); torch.cuda.synchronize()
    sig = [o["class_logits"].double().sum().item()] + [o[s][k].double().sum().item() for s in ("left", "right") for k in ("vertices", "j3d", "transl")]
    sig.append(int(net.net.debug_buffer("fps1", torch.int32).long().sum())); sig.append(int(net.net.debug_buffer("gidx1_2", torch.int32).long().sum()))
    for _ in range(5): once()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k = 40
    for _ in range(k): once()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / k
    g = net.net.capture(x, net.hands, inits)
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): g.replay()
    torch.cuda.synchronize(); dtg = (time.perf_counter() - t0) / k
    out[f"{B}x{N}"] = {"sig": sig, "eager_ms": round(dt * 1e3, 4), "graph_ms": round(dtg * 1e3, 4), "win_s": round(B / dt), "win_s_graph": round(B / dtg)}
import json; print("RESULT " + json.dumps(out))
''' % ROOT
res = {}
for prec in sys.argv[1:] or ["f16x2"]:
    for tag, env in (("chunked", {"EV2H_FPS_CHUNKS": os.environ.get("CHUNK_ON", "")} if os.environ.get("CHUNK_ON") else {}), ("one launch", {"EV2H_FPS_CHUNKS": "0"})):
        e = {k: v for k, v in os.environ.items() if k != "EV2H_FPS_CHUNKS"}
        r = subprocess.run([sys.executable, "-c", code, prec], env=dict(e, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
        assert line, r.stderr[-2000:]
        res[tag] = json.loads(line[0][7:])
    for shape in res["chunked"]:
        a, b = res["chunked"][shape], res["one launch"][shape]
        print(f"{prec} {shape:8s} identical {a['sig'] == b['sig']}  eager {b['eager_ms']:.3f} -> {a['eager_ms']:.3f} ms ({b['win_s']} -> {a['win_s']} windows/s)  hipGraph {b['graph_ms']:.3f} -> {a['graph_ms']:.3f} ms ({b['win_s_graph']} -> {a['win_s_graph']})")
