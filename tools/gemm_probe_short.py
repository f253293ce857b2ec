import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/manuscript_ocr_amd") else os.getcwd())
sys.path.insert(0, os.getcwd())
from tools.gemm_probe import run
for (M, N, K) in ((1572864, 256, 64), (1572864, 128, 64), (1572864, 64, 64), (1572864, 512, 64), (393216, 512, 128), (1572864, 64, 256), (98304, 1024, 256),
                  (24 * 204800, 128, 64), (24 * 204800, 128, 128)):
    run(M, N, K)
