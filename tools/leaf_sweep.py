import sys; sys.path.insert(0,'tools'); sys.path.insert(0,'.')
import quick_perf as q
for leaf in (1,2,3,4,2,1):
    q.run(1920, 128, 50, q.pkg.ACCEL_BVH, leaf=leaf, reps=3)
