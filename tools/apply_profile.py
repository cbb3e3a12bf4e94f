import sys, cProfile, pstats
sys.path.insert(0, ".")
exec(open("tools/displacement_timing.py").read().split("for _ in range(3):")[0])
reg.apply(img)
pr = cProfile.Profile(); pr.enable(); reg.apply(img); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
