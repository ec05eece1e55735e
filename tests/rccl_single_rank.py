"""Single-rank RCCL ("nccl" backend) run of the device-buffer branch of geneevolve_amd/distributed.py on the one GPU of the test
box: see tests/test_gpu_parity.py::test_rccl_backend_single_rank_collectives_drive_the_device_buffer_branch."""
import os
import socket
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.distributed as dist
    from geneevolve_amd.capi import GevLibrary
    from geneevolve_amd.distributed import compute_ad_locus_split, migrate_all_to_all
    from geneevolve_amd.host import Simulation, SyntheticConfig, synthetic_random_mate
    from tests import helpers
    print("torch sees", torch.cuda.device_count(), "device(s)", flush=True)
    torch.cuda.set_device(0)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    cfg = SyntheticConfig(200, 3000, nchr=2, chrom_bp=1_000_000, map_step=1000, rec_per_row=3e-3, mut_per_row=3e-3, n_cv=50, seed=4)
    g = GevLibrary().create(1, 2, 1, 0)
    cfg.apply_static(g)
    for c in range(2):
        g.synth_founders(0, c, 400, 10 + c); g.synth_cv_founders(0, 0, c, 400, 20 + c)
    sim = Simulation(g, 5, 2, True)
    sim.ras_initial_human_gen0(0, 200)
    sim.couples[0] = synthetic_random_mate(sim.sex[0], 200, np.random.default_rng(1)); sim.reproduce(0, 1)
    before = [g.download_haps(0, c) for c in range(2)]
    add0, dom0, addc0, domc0 = g.compute_ad(0)
    trace = []
    got = migrate_all_to_all(g, [np.empty(0, dtype=np.uint64)], 0, device="cuda:0", trace=trace)
    assert got == [0] and "a2a_meta" in trace and trace.count("fence") == 2, trace
    add, dom, addc, domc = compute_ad_locus_split(g, 0, device="cuda:0")        # all_reduce over RCCL IN PLACE on the library's own device arrays
    assert helpers.bits_equal(add, add0) and helpers.bits_equal(addc, addc0) and helpers.bits_equal(dom, dom0) and helpers.bits_equal(domc, domc0)
    pa, pd, n = g.compute_ad_device(0)                                          # (the pointers are the library's; a second reduction over one rank changes nothing)
    assert n == 200 * 2 * 1 and pa and pd
    add2, dom2, addc2, domc2 = g.ad_finish_device(0)
    assert helpers.bits_equal(add2, add0) and helpers.bits_equal(addc2, addc0)
    for c in range(2):
        assert np.array_equal(g.download_haps(0, c), before[c])
    g.close()
    dist.barrier()
    dist.destroy_process_group()
    print("rccl single rank ok")


if __name__ == "__main__":
    main()
