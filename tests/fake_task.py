"""Trivial task 'image' for lifecycle tests: argv = the entry contract; the
`file` field selects the behaviour: sleep:<s>, exit:<code>, or ps tasks join forever."""
import signal
import sys
import time


def main():
    job_name, idx, ps, workers, file_field = sys.argv[1:6]
    print("fake task", job_name, idx, file_field, flush=True)
    if job_name == "ps":
        if "ignore-term" in file_field:
            signal.signal(signal.SIGTERM, signal.SIG_IGN)
        while True:
            time.sleep(0.1)
    kind, _, arg = file_field.partition(":")
    if kind == "sleep":
        time.sleep(float(arg.split(",")[0]))
        return 0
    if kind == "exit":
        return int(arg)
    if kind == "failone":
        # worker 1 fails, the others hang (as if blocked in a collective)
        if idx == "1":
            time.sleep(0.2)
            return 3
        while True:
            time.sleep(0.1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
