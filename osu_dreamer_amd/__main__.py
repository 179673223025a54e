from .fit import main

main()
