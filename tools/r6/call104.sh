#!/bin/bash
# the driver's round-end commands on the final tree (library rebuilt from the final sources): whole GPU suite, smoke
mkdir -p gpurun_out/r6
( time python -m pytest tests/ -x -q -m gpu ) > gpurun_out/r6/suite_driver_form_call104.log 2>&1; tail -5 gpurun_out/r6/suite_driver_form_call104.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
