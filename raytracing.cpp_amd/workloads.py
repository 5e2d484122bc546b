"""Synthetic workloads of BASELINE.json's configs: scene + camera generators shared by bench.py and the tests.

Config 1: three Lambertian spheres (plumbing).  Configs 2/3: S-RTOW = make_world_spheres (the reference's generator,
core.cc:99-149) under the camera of data/config/world.config.json.  Config 4: 100k spheres on a jittered grid.
Config 5: a Cornell-like box made of huge spheres.  Pure numpy; nothing here touches a device."""
import numpy as np

OBJECT_DTYPE = np.dtype([("kind", "<u4"), ("center", "<f4", 3), ("radius", "<f4"), ("material", "<u4")])
MATERIAL_DTYPE = np.dtype([("kind", "<u4"), ("p", "<f4", 4)])


def arrays(spec):
    """spec: list of (center, radius, (kind, p4)); one material per object."""
    objs = np.zeros(len(spec), OBJECT_DTYPE)
    mats = np.zeros(len(spec), MATERIAL_DTYPE)
    for i, (c, r, (k, p)) in enumerate(spec):
        objs[i] = (0, c, r, i)
        mats[i] = (k, p)
    return objs, mats


def three_spheres():
    """BASELINE config 1: ground + two Lambertian spheres."""
    return arrays([((0.0, -100.5, -1.0), 100.0, (0, (0.8, 0.8, 0.0, 0.0))),
                   ((-0.6, 0.0, -1.0), 0.5, (0, (0.1, 0.2, 0.5, 0.0))),
                   ((0.6, 0.0, -1.0), 0.5, (0, (0.8, 0.3, 0.3, 0.0)))])


def three_spheres_camera():
    return dict(aspect_ratio=16.0 / 9.0, image_width=400, samples_per_pixel=1, max_depth=1, vertical_fov=90.0,
                defocus_angle=0.0, focus_distance=1.0, lookfrom=(0.0, 0.0, 0.0), lookat=(0.0, 0.0, -1.0),
                world_up=(0.0, 1.0, 0.0))


def cornell_like():
    """BASELINE config 5 shape: a box made of huge Lambertian spheres (R = 1e3) -- floor, ceiling, red and green side
    walls, a bright back wall -- open towards the camera so that sky light enters (the reference has no emitters),
    one glass and one metal sphere inside.  Returns (objs, mats, camera kwargs)."""
    R, h = 1.0e3, 5.0  # fp32 self-intersection swallows most paths at R = 1e4 (reference arithmetic); 1e3 keeps signal
    spec = [
        ((0.0, -R - h, 0.0), R, (0, (0.73, 0.73, 0.73, 0.0))),   # floor
        ((0.0, R + h, 0.0), R, (0, (0.73, 0.73, 0.73, 0.0))),    # ceiling
        ((-R - h, 0.0, 0.0), R, (0, (0.65, 0.05, 0.05, 0.0))),   # left, red
        ((R + h, 0.0, 0.0), R, (0, (0.12, 0.45, 0.15, 0.0))),    # right, green
        ((0.0, 0.0, -R - h), R, (0, (0.99, 0.99, 0.99, 0.0))),   # back, bright
        ((-2.0, -3.0, -1.0), 2.0, (2, (1.5, 0.0, 0.0, 0.0))),
        ((2.2, -3.2, 1.0), 1.8, (1, (0.8, 0.85, 0.88, 0.05))),
    ]
    objs, mats = arrays(spec)
    cam = dict(aspect_ratio=1.0, image_width=800, samples_per_pixel=4096, max_depth=200, vertical_fov=40.0,
               defocus_angle=0.0, focus_distance=10.0, lookfrom=(0.0, 0.0, 18.0), lookat=(0.0, 0.0, 0.0),
               world_up=(0.0, 1.0, 0.0))
    return objs, mats, cam


def random_spheres(n, seed=1, extent=10.0):
    """n random spheres of mixed radii and materials over a ground sphere (BVH stress)."""
    rng = np.random.default_rng(seed)
    spec = [((0.0, -1000.0, 0.0), 1000.0, (0, (0.5, 0.5, 0.5, 0.0)))]
    for _ in range(n):
        r = float(rng.choice([0.1, 0.2, 0.35, 0.8]))
        c = (float(rng.uniform(-extent, extent)), r + float(rng.uniform(0, 0.5)), float(rng.uniform(-extent, extent)))
        k = int(rng.choice([0, 0, 0, 1, 2]))
        if k == 0:
            p = (*[float(v) for v in rng.uniform(0.05, 0.95, 3)], 0.0)
        elif k == 1:
            p = (*[float(v) for v in rng.uniform(0.5, 1.0, 3)], float(rng.uniform(0, 0.5)))
        else:
            p = (float(rng.uniform(1.2, 1.7)), 0.0, 0.0, 0.0)
        spec.append((c, r, (k, p)))
    return arrays(spec)


def big_grid(n_side=316, seed=4):
    """BASELINE config 4 shape: n_side^2 spheres (R = 0.2) on a jittered unit grid over a ground sphere, RTOW material
    mix (80 % Lambertian, 15 % Metal, 5 % Dielectric); returns (objs, mats, camera kwargs)."""
    rng = np.random.default_rng(seed)
    n = n_side * n_side
    objs = np.zeros(n + 1, OBJECT_DTYPE)
    mats = np.zeros(n + 1, MATERIAL_DTYPE)
    half = n_side / 2.0
    objs[0] = (0, (0.0, -10000.0, 0.0), 10000.0, 0)
    mats[0] = (0, (0.5, 0.5, 0.5, 0.0))
    gx, gz = np.meshgrid(np.arange(n_side), np.arange(n_side), indexing="ij")
    cx = (gx.ravel() - half + 0.9 * rng.random(n)).astype(np.float32)
    cz = (gz.ravel() - half + 0.9 * rng.random(n)).astype(np.float32)
    objs["center"][1:, 0] = cx
    objs["center"][1:, 1] = 0.2
    objs["center"][1:, 2] = cz
    objs["radius"][1:] = 0.2
    objs["material"][1:] = np.arange(1, n + 1)
    choose = rng.random(n)
    kind = np.where(choose < 0.8, 0, np.where(choose < 0.95, 1, 2)).astype(np.uint32)
    mats["kind"][1:] = kind
    p = np.zeros((n, 4), np.float32)
    lam, met, die = kind == 0, kind == 1, kind == 2
    p[lam, :3] = (rng.random((lam.sum(), 3)) * rng.random((lam.sum(), 3))).astype(np.float32)
    p[met, :3] = rng.uniform(0.5, 1.0, (met.sum(), 3)).astype(np.float32)
    p[met, 3] = rng.uniform(0.0, 0.5, met.sum()).astype(np.float32)
    p[die, 0] = rng.uniform(1.2, 1.6, die.sum()).astype(np.float32)
    mats["p"][1:] = p
    scale = n_side / 22.0
    cam = dict(aspect_ratio=16.0 / 9.0, image_width=1920, samples_per_pixel=256, max_depth=50, vertical_fov=20.0,
               defocus_angle=0.6, focus_distance=10.0 * scale, lookfrom=(13.0 * scale, 2.0 * scale, 3.0 * scale),
               lookat=(0.0, 0.0, 0.0), world_up=(0.0, 1.0, 0.0))
    return objs, mats, cam


def fuzz_world(rng, case):
    """One random world of the differential fuzz (tools/fuzz_vs_oracle.py, tests/test_gpu_parity.py): 1-90 spheres -- every
    fourth world 100-600 spheres over a field hundreds of radii wide, where the reach-bounded box pad is the library's choice --
    negative radii, scales 1e-2 .. 1e3, all three materials, bounce limits 1-120, defocus on and off, half of the worlds over a
    huge ground sphere.  Returns (objs, mats, camera kwargs); `rng` is a numpy Generator, `case` selects the wide worlds."""
    wide = case % 4 == 3
    n = int(rng.integers(100, 600)) if wide else int(rng.integers(1, 90))
    n_mats = int(rng.integers(1, n + 1))
    objs = np.zeros(n, OBJECT_DTYPE)
    mats = np.zeros(n_mats, MATERIAL_DTYPE)
    scale = float(10.0 ** rng.uniform(-2, 3))
    for m in range(n_mats):
        k = int(rng.integers(0, 3))
        if k == 0:
            mats[m] = (0, (*rng.uniform(0.0, 1.2, 3), 0.0))
        elif k == 1:
            mats[m] = (1, (*rng.uniform(0.3, 1.0, 3), float(rng.uniform(0.0, 1.5))))
        else:
            mats[m] = (2, (float(rng.uniform(0.6, 2.2)), 0, 0, 0))
    objs["center"] = (rng.normal(0, 3.0, (n, 3)) * scale).astype(np.float32)
    objs["radius"] = (10.0 ** rng.uniform(-1.5, 0.8, n) * scale * rng.choice([1.0, 1.0, 1.0, -1.0], n)).astype(np.float32)
    if wide:  # a field hundreds of radii wide, flat or not
        objs["center"] = (rng.uniform(-1.0, 1.0, (n, 3)) * (float(rng.uniform(30, 300)), float(rng.choice([0.5, 30.0])),
                                                           float(rng.uniform(30, 300))) * scale).astype(np.float32)
        objs["radius"] = (10.0 ** rng.uniform(-1.0, 0.0, n) * scale).astype(np.float32)
    objs["material"] = rng.integers(0, n_mats, n)
    if rng.random() < 0.5:
        objs["center"][0] = (0, -1000.0 * scale - scale, 0)
        objs["radius"][0] = 1000.0 * scale
    depth = int(rng.choice([1, 3, 8, 20, 50, 120]))
    lf = tuple(float(v) for v in rng.normal(0, 6.0, 3) * scale)
    kw = dict(aspect_ratio=1.0, image_width=int(rng.choice([17, 32, 40])), samples_per_pixel=int(rng.choice([1, 4, 9])),
              max_depth=depth, vertical_fov=float(rng.uniform(20, 90)), defocus_angle=float(rng.choice([0.0, 0.5, 3.0])),
              focus_distance=float(5 * scale), lookfrom=lf, lookat=(0.0, 0.0, 0.0), world_up=(0.0, 1.0, 0.0))
    return objs, mats, kw
