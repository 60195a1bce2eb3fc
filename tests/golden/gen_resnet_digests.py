#!/usr/bin/env python3
"""Per-layer digests of the ORACLE's encrypted ResNet (tests/oracle_resnet.py: testResNet_crop_sparse, test.go:76-370, wide_case 1, synthetic weights) at full size
(N = 2^16): the SHA-256 of the ciphertext every conv-BN-ReLU layer hands on, and the class scores. The oracle network takes about 3 minutes per layer on one core, so it
is run once here (build container) and the GPU test (tests/test_gpu_a_parity.py::test_resnet_network_on_gpu) compares the device network with these digests.

    python tests/golden/gen_resnet_digests.py 8 20        # depths; default: 8
"""
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_resnet as orn  # noqa: E402
import parity_cases as pc  # noqa: E402

if __name__ == "__main__":
    depths = [int(a) for a in sys.argv[1:]] or [8]
    path = pc.RESNET_FIXTURE
    for depth in depths:
        doc = json.load(open(path)) if os.path.exists(path) else {"logN": 16, "made_by": "tests/golden/gen_resnet_digests.py", "depth": {}}
        t0 = time.time()
        layers, scores = pc.resnet_layer_digests(orn.ResNetOracle(orn.Net(16, depth=depth)))
        doc = json.load(open(path)) if os.path.exists(path) else doc          # another depth may have finished meanwhile
        doc["depth"][str(depth)] = {"layers": layers, "scores": scores}
        json.dump(doc, open(path, "w"), indent=1, sort_keys=True)
        print("depth", depth, f"{time.time() - t0:.0f} s", scores, flush=True)
