"""Every rider of the bench in ONE process, so that one `rocprofv3 --kernel-trace --stats -- python tools/riders.py` gives the
kernel summary of all of them: the other BASELINE configs (n=2 searches, config 3, the config-5 masked scorer), the materialised
generators (n=3 burst, n=2 render), the device-resident chain (generator -> plain scorer -> solve_batch) and the host-buffer
batch operators.  Prints one JSON object {tool: its JSON}."""
import io
import json
import os
import runpy
import sys
from contextlib import redirect_stdout

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}
for tool, argv in (("bench_configs.py", []), ("enum_profile.py", []), ("device_chain.py", ["26"]), ("batch_profile.py", [])):
    buf = io.StringIO()
    old = sys.argv
    sys.argv = [os.path.join(HERE, tool)] + argv
    try:
        with redirect_stdout(buf):
            runpy.run_path(os.path.join(HERE, tool), run_name="__main__")
        out[tool] = json.loads(buf.getvalue().strip().splitlines()[-1])
    except BaseException as e:          # a tool that fails must not take the others' numbers with it
        out[tool] = {"error": repr(e), "stdout": buf.getvalue()[-500:]}
    finally:
        sys.argv = old
print(json.dumps(out))
