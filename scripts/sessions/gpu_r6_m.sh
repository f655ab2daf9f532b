#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
/opt/rocm/bin/hipcc -O2 -o /tmp/pcie_2d scripts/probe/pcie_2d.cpp && timeout 300 /tmp/pcie_2d > gpurun_out/r06m_pcie_2d.txt 2>&1; cat gpurun_out/r06m_pcie_2d.txt
