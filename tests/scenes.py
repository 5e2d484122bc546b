"""Synthetic scenes shared by the tests: re-exported from the package's workload generators."""
import rtmi_loader

_w = rtmi_loader.load().workloads
arrays, three_spheres, three_spheres_camera = _w.arrays, _w.three_spheres, _w.three_spheres_camera
cornell_like, random_spheres, big_grid = _w.cornell_like, _w.random_spheres, _w.big_grid
