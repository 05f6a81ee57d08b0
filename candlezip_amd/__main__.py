from candlezip_amd.cli import main

main()
