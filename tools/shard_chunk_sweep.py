"""Work-item sizes against the trace-kernel time of the shard rank 0 of a G-GPU job renders (S-RTOW 1080p x 512 spp, or another spp):
the library's own choice (~128 items per lane, >= 4 samples) against fixed sizes.  (profiles/r04_two_phase_chunks.txt also holds the
round-4 experiment with two phases -- long chunks, then a tail of chunks of 3 -- which this tool drove while that code existed.)
usage: shard_chunk_sweep.py [spp] [G,G,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, rtmi_loader
pkg = rtmi_loader.load()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
gs = [int(g) for g in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
objs, mats = pkg.make_world_spheres(12345)
dev = torch.device("cuda:0"); stream = torch.cuda.current_stream(dev).cuda_stream
cam = pkg.camera_setup(pkg.camera_params(image_width=1920, samples_per_pixel=spp, max_depth=50))
W, H = cam.img_width, cam.img_height
for G in gs:
    plan = pkg.RowShardPlan(H, 8, G)
    y_first, n_blocks, rows = plan.shard(0)
    buf = torch.zeros((plan.max_rows, W, 3), dtype=torch.float32, device=dev)
    with pkg.Scene(cam, objs, mats) as sc:
        li = sc.launch_info()
    lanes, pixels = li["grid_blocks"] * li["block_lanes"], rows * W
    n_chunks = min(spp, -(-128 * lanes // pixels))
    old = max(3, -(-spp // n_chunks))
    variants = [(f"auto (rule: {max(4, old)})", None)]
    for c in (2, 3, 4, 5, 6, 8, 12, 16, 24, 32):
        if c < spp:
            variants.append((f"chunks of {c}", dict(chunk_samples=c)))
    for name, tun in variants:
        with pkg.Scene(cam, objs, mats, tuning=tun) as sc:
            ms = []
            for _ in range(3):
                sc.render_row_blocks_device(y_first, 8, G, n_blocks, 7, buf.data_ptr(), 0, stream)
                torch.cuda.synchronize(dev)
                ms.append(sc.last_kernel_ms())
        print(f"G={G} {spp} spp  {name:36s} {min(ms):8.2f} ms", flush=True)
