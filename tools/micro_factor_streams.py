"""Do factorisations on different HIP streams overlap?  ms per 4096 x 4096 factorisation with 1..8 streams."""

import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from sleekit_amd import _device as dev  # noqa: E402
from sleekit_amd import codebook, synth  # noqa: E402
from sleekit_amd import dist as sdist  # noqa: E402


def main():
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dev.lazy_errors = True
    n = int(os.environ.get("N", "4096"))
    cb = codebook.UniformCodebook(8, -1, 1)
    lay = synth.make_layer(64, n, 1000, device=device)
    layer = {k: torch.from_numpy(lay[k]).to(device) for k in ("W", "H", "scale")}
    backend = sdist.HipBackend(cb, "diag", 0.01, 0, with_error=True, overlap=False)
    backend.factorize(layer)
    torch.cuda.synchronize()
    for ns in (1, 2, 3, 4, 6, 8):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        count = 24

        def run():
            for i in range(count):
                with torch.cuda.stream(streams[i % ns]):
                    backend.factorize(layer)

        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
        print(f"{ns} streams: {1e3 * tot / count:7.3f} ms per factorisation (host {1e3 * host / count:6.3f})", flush=True)


if __name__ == "__main__":
    main()
