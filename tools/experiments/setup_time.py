import sys, time, torch
sys.path.insert(0, '.')
import bench
sys.argv = ['bench.py']
args = bench.parse()
dev = torch.device('cuda')
t0 = time.time(); data, model, neg, ni1, ni2 = bench.build_request(args, dev); torch.cuda.synchronize(); print('build_request (synthetic data + masks)', round(time.time() - t0, 2), 's')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.time(); eng = bench.make_engine(args, data, model, neg, ni1, ni2, dev); torch.cuda.synchronize(); t1 = time.time()
eng.step(); torch.cuda.synchronize(); t2 = time.time()
pr.disable()
print('engine construction', round(t1 - t0, 3), 's; first step incl. graph capture', round(t2 - t1, 3), 's')
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
