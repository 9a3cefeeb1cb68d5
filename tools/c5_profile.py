"""host-side profile of one config-5 volume through pipe_gray3d_slic_features_model_graphcut (cProfile, cumulative)"""
import cProfile, pstats, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from pyimsegm_amd import pipelines as pipe, _hip
from pyimsegm_amd.utilities.synthetic import config5_volume
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '64,4096,4096').split(','))
vol = config5_volume(shape, seed=5)
p = bench.C5_PARAMS
def step():
    np.random.seed(0)
    return pipe.pipe_gray3d_slic_features_model_graphcut(vol, bench.NB_CLASSES, {'color': ('mean', 'std', 'energy')}, spacing=p['spacing'],
                                                         sp_size=p['sp_size'], sp_regul=p['sp_regul'], gc_regul=p['gc_regul'])
t = time.perf_counter(); step(); print('warm-up step %.2f s' % (time.perf_counter() - t))
pr = cProfile.Profile()
t = time.perf_counter(); pr.enable(); step(); pr.disable(); print('profiled step %.2f s' % (time.perf_counter() - t))
st = pstats.Stats(pr); st.sort_stats('cumulative').print_stats(45)
