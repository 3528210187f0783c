import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import exp_conv as E
E.run_wino(16, 64, 128, 128, 1, 0, reps=3)
