#!/usr/bin/env python3
"""Reads ThreadSanitizer output of tools/tsan_cli.sh and keeps the reports in which NEITHER access runs through the
uninstrumented HIP / HSA runtime (whose internal locking TSan cannot see: objects the runtime allocates on one of its
threads and frees on another are reported as races on every run)."""
import re, sys
for path in sys.argv[1:]:
    text = open(path, errors="replace").read()
    reports = re.split(r"(?m)^={18}\n", text)
    kept = 0
    for r in reports:
        if "WARNING: ThreadSanitizer" not in r:
            continue
        # the two access stacks: from "Write of/Read of/Previous ..." up to the blank line after each
        stacks = re.findall(r"(?ms)^  (?:Write|Read|Previous|Atomic)[^\n]*\n((?:    #\d+[^\n]*\n)+)", r)
        runtime = [bool(re.search(r"libamdhip64|libhsa-runtime|libhsakmt|librocprofiler|libamd_comgr", s.split("\n")[1] if len(s.split("\n")) > 1 else s) or
                        re.search(r"libamdhip64|libhsa-runtime", "\n".join(s.split("\n")[:3]))) for s in stacks]
        if stacks and not any(runtime):
            kept += 1
            print(f"---- {path}\n{r.strip()[:3000]}\n")
    print(f"{path}: {len([r for r in reports if 'WARNING: ThreadSanitizer' in r])} reports, {kept} outside the HIP/HSA runtime")
