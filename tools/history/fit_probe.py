import sys, time, warnings, os
sys.path.insert(0, '/root/repo')
warnings.filterwarnings('ignore')
import numpy as np
from threadpoolctl import threadpool_limits, threadpool_info
from sklearn import mixture
from pyimsegm_amd import graph_cuts as g
rng=np.random.default_rng(0)
K=298116
X=np.vstack([rng.normal([0,0,0],[1,.5,.8],(K//3,3)), rng.normal([3,1,2],[.7,.6,.5],(K//3,3)), rng.normal([-2,2,1],[.5,.9,.6],(K-2*(K//3),3))])
X += rng.normal(0, 1.5, X.shape)
print('cpus', os.cpu_count(), [(p['user_api'], p['num_threads']) for p in threadpool_info()])
for limit in (32, 64, None):
    ctx = threadpool_limits(limits=limit) if limit else None
    for rep in range(2):
        np.random.seed(3); t=time.time(); b=g.fit_mixture_restarts(mixture.GaussianMixture(3, covariance_type='full', n_init=9, max_iter=99), X); tb=time.time()-t
        print('limit', limit, 'side by side %.3f s' % tb, 'iters', b.n_iter_)
    np.random.seed(3); t=time.time(); a=mixture.GaussianMixture(3, covariance_type='full', n_init=9, max_iter=99).fit(X); ta=time.time()-t
    print('limit', limit, 'plain fit %.3f s' % ta, 'equal', all(np.array_equal(getattr(a,n), getattr(b,n)) for n in ('weights_','means_','covariances_','precisions_cholesky_')))
    # one restart alone: init and EM
    m=mixture.GaussianMixture(3, covariance_type='full', n_init=1, max_iter=99); m._check_parameters(X)
    rs=np.random.RandomState(1); t=time.time(); m._initialize_parameters(X, rs); ti=time.time()-t
    t=time.time(); lp,lr=m._e_step(X); te=time.time()-t; t=time.time(); m._m_step(X,lr); tm=time.time()-t
    print('   one restart alone: init %.3f  e-step %.3f  m-step %.3f' % (ti, te, tm))
    if ctx: ctx.restore_original_limits()
