"""Developer A/B: run the given tool scripts once per library build (DARSIA_HIP_LIB), printing their output."""
import os, subprocess, sys
libs, scripts = sys.argv[1].split(","), sys.argv[2:]
for sc in scripts:
    for lib in libs:
        env = dict(os.environ)
        if lib != "default": env["DARSIA_HIP_LIB"] = os.path.abspath(lib)
        out = subprocess.run([sys.executable] + sc.split(), env=env, capture_output=True, text=True)
        print(f"--- {sc} [{lib}]\n{out.stdout.strip()[-1500:]}{out.stderr.strip()[-300:] if out.returncode else ''}", flush=True)
