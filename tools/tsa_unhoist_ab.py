"""Does TemporalSelfAttention's sampling run faster on values projected right before it (hot in the memory-side cache) than on
the values the hoisted grouped GEMM wrote at the start of the frame?  Base frame, graph replay: the default schedule against one
where only the camera values are hoisted and every layer projects its own TSA value (N = 256)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import time
import torch
import bevformer_amd
from bevformer_amd import synthetic as S

DEV = torch.device("cuda:0")


def main():
    torch.manual_seed(0)
    enc = bevformer_amd.build_transformer_layer_sequence(S.encoder_cfg("base")).eval()
    sd = S.trained_like_({k: v.clone() for k, v in enc.state_dict().items()}, seed=3)
    enc.load_state_dict(sd)
    enc = enc.to(DEV)
    q, f, kw = S.make_inputs("base", seed=0, temporal=True, device=DEV)
    real = enc.hoisted_value_projections

    def only_sca(value, tsa_value, **k):
        sca, _ = real(value, None, **k)
        return sca, None

    def timed(label):
        with torch.no_grad():
            for _ in range(3):
                out = enc(q, f, f, **kw)
            torch.cuda.synchronize()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                enc(q, f, f, **kw)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = enc(q, f, f, **kw)
            g.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(20):
                    g.replay()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / 20 * 1e3)
        print(f"{label}: ms per frame {sorted(ts)[2]:.4f} (windows {[round(t, 4) for t in ts]})")
        return out.clone()

    for r in range(2):
        enc.hoisted_value_projections = real
        a = timed("hoisted TSA values (default)")
        enc.hoisted_value_projections = only_sca
        b = timed("per-layer TSA value projection ")
        print("   max abs difference of the outputs", float((a - b).abs().max()))


main()
