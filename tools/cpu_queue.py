#!/usr/bin/env python
"""Background CPU job queue for the reference-module fixture runs (build container only).

    python tools/cpu_queue.py QUEUE_FILE [-P 3]

Runs the shell commands of QUEUE_FILE (one per line, '#' comments; lines may be APPENDED while it runs) at most P at a
time, each with stdout/stderr in QUEUE_FILE.logs/<n>.log; exits when the file holds the line `__END__` and all jobs are done.
"""
import argparse
import os
import subprocess
import sys
import time

ap = argparse.ArgumentParser()
ap.add_argument("queue")
ap.add_argument("-P", type=int, default=3)
a = ap.parse_args()
logs = a.queue + ".logs"
os.makedirs(logs, exist_ok=True)
started, running = 0, {}
while True:
    lines = [l.strip() for l in open(a.queue) if l.strip() and not l.startswith("#")]
    end = "__END__" in lines
    jobs = [l for l in lines if l != "__END__"]
    for n in [n for n, p in running.items() if p.poll() is not None]:
        print("job %d done rc=%d" % (n, running.pop(n).returncode), flush=True)
    while started < len(jobs) and len(running) < a.P:
        f = open(os.path.join(logs, "%03d.log" % started), "w")
        f.write("$ " + jobs[started] + "\n"); f.flush()
        running[started] = subprocess.Popen(jobs[started], shell=True, stdout=f, stderr=subprocess.STDOUT, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        print("job %d started: %s" % (started, jobs[started]), flush=True)
        started += 1
    if end and not running and started >= len(jobs):
        break
    time.sleep(5)
