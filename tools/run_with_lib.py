"""A/B helper: run a script with another build of the library.  usage: run_with_lib.py <lib.so> <script.py> [args...]"""
import sys, runpy, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asr_amd._lib as L
L.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
