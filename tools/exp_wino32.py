import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_conv as E
for (cin, cout, h, glu, res) in ((32, 32, 128, 0, 0), (32, 32, 64, 0, 0), (32, 32, 32, 0, 1)):
    E.run(16, cin, cout, h, glu, 0, res)
    E.run_wino(16, cin, cout, h, glu, res)
