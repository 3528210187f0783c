import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_conv as E
for (cin, cout, h, glu, res) in ((64, 128, 128, 1, 0), (64, 64, 128, 0, 1), (64, 128, 64, 1, 0), (64, 64, 64, 0, 1), (32, 64, 128, 1, 0), (32, 64, 32, 1, 0), (64, 128, 32, 1, 0)):
    E.run(16, cin, cout, h, glu, 0, res)
    E.run_wino(16, cin, cout, h, glu, res)
