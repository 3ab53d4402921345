import cProfile, pstats, sys
sys.argv=['x','--members',sys.argv[1] if len(sys.argv)>1 else '32','--iterations',sys.argv[2] if len(sys.argv)>2 else '4']
sys.path.insert(0,'examples')
import multi_input_ensemble as M
cProfile.run('M.main()','/tmp/prof.out')
p=pstats.Stats('/tmp/prof.out'); p.sort_stats('cumulative').print_stats('difflexmm_amd|examples', 30)
